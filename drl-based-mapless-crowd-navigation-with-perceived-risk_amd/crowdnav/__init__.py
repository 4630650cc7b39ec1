"""crowdnav -- MI355X-native batched crowd-navigation environment (host side).

The compute lives in lib/libcrowdnav.so (hand-written HIP, gfx950); this package is the Python
mirror of the reference's `Env` interface over its C-ABI."""
from .config import Config  # noqa: F401
from ._abi import CrowdNavError, build, lib  # noqa: F401


def __getattr__(name):
    if name in ("VecEnv", "VecEnvGroups", "Env"):
        from . import env
        return getattr(env, name)
    raise AttributeError(name)
