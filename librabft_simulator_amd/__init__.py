"""MI355X-native batched LibraBFTv2 discrete-event simulator (HIP, gfx950).

Drop-in for the simulation hot path of novifinancial/librabft_simulator behind the C ABI of
include/lbft.h.  The package contains only what that path needs: csrc/ (hand-written HIP kernels +
the C ABI), the ctypes binding and the host-side mirror of the reference's simulator interface.
"""
from ._lib import LbftError, lib  # noqa: F401
from .simulator import (BatchResult, BatchSimulator, Command, Duration, GlobalTime, NodeConfig, NodeTime,  # noqa: F401
                        RandomDelay, Simulator, State)

__all__ = ["BatchSimulator", "BatchResult", "Simulator", "RandomDelay", "NodeConfig", "GlobalTime", "NodeTime", "Duration", "State",
           "Command", "LbftError", "lib"]
