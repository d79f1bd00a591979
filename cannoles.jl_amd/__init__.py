"""MI355X-native Newton-system backend for CaNNOLeS."""
from . import sharding, synthetic  # noqa: F401
