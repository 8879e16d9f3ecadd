"""svdss_amd -- MI355X-native hot path of Parsoa/SVDSS (`search`: ping-pong SFS extraction).

The compute lives in libsvdss_hip.so (hand-written HIP for gfx950 behind the
C-ABI of include/svdss_hip.h); this package is the thin host mirror of the
reference's interface for that path.  Importing it without the built library
raises ImportError -- there is no CPU fallback.
"""
from ._lib import LIB_PATH, SVDSS_SFS_ASSEMBLE, SvdssError, lib  # noqa: F401
from .pingpong import (FMDIndex, NT6_TABLE, PingPong, SFSBatch, nt6_encode, output_batch,  # noqa: F401
                       pack_reads, parse_sfsfile)
from . import calldp  # noqa: F401,E402
