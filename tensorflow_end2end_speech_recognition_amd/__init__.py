"""MI355X-native BLSTM/VGG-BLSTM -> CTC / attention training + decode path.

Host-side mirror of the hot-path class surface of
hirofumi0810/tensorflow_end2end_speech_recognition (models/, utils/ keep the
reference's module paths) over hand-written gfx950 kernels behind the C ABI in
include/asr_hip.h (libasr_hip.so, built in-tree by `build.py`).
"""
__version__ = '0.1.0'
