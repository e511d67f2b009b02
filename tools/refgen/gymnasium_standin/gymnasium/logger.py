import warnings
def warn(msg, *a): warnings.warn(msg % a if a else msg)
