"""avvad -- Python host of the MI355X-native AV-VAD hot path (ctypes over libavvad_hip.so)."""
from ._lib import AvvadError, LIB_PATH, lib  # noqa: F401
