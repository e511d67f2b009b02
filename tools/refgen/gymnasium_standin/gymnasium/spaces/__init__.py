from .space import Space
from .box import Box
from .discrete import Discrete
from .multi_discrete import MultiDiscrete
from .tuple import Tuple
