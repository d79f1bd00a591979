"""MI355X-native Newton-system backend for CaNNOLeS."""
from . import batch_solve, outer_loop, sharding, synthetic  # noqa: F401
