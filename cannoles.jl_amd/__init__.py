"""MI355X-native Newton-system backend for CaNNOLeS."""
from . import synthetic  # noqa: F401
