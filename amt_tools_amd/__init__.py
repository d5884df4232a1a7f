"""
amt_tools_amd -- MI355X-native (gfx950) implementation of amt-tools' frame-level transcription hot
path behind amt-tools' own FeatureModule / TranscriptionModel plugin API.

    from amt_tools_amd.features import MelSpec, STFT, CQT, HCQT        # amt_tools.features mirror
    from amt_tools_amd.models import OnsetsFrames, OnsetsFrames2       # amt_tools.models mirror

The arithmetic lives in hand-written HIP kernels (amt_tools_amd/csrc, C-ABI declared in
include/amtx.h) loaded through ctypes; PyTorch is used for device memory, streams and
torch.distributed only.  See DESIGN.md.
"""

__version__ = '0.1.0'
