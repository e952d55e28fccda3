"""visiondk_amd — MI355X-native (gfx950) hot path of wuji3/visiondk behind the reference's plugin surface.

Only what the path needs lives here: `csrc/` (hand-written HIP kernels + the C ABI of include/visiondk.h) and
the host-side mirror of the reference interfaces (model factory, step protocol, retrieval index).
"""
__version__ = "0.1.0"
