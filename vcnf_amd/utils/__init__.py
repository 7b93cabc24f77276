from . import masks, splines                                     # noqa: F401
from .masks import (create_alternating_binary_mask, create_mid_split_binary_mask,   # noqa: F401
                    create_random_binary_mask)
from .nn import sum_except_batch                                 # noqa: F401
