"""Test-side name for the ctypes binding of the C++ host mirror."""
from stark_amd.hostlib import *          # noqa: F401,F403
from stark_amd.hostlib import lib, _vp   # noqa: F401
