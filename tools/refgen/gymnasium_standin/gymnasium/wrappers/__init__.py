from .. import error
def __getattr__(name):
    raise error.DependencyNotInstalled("gymnasium stand-in: wrappers." + name + " unavailable")
