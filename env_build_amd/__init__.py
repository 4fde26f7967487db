"""env_build_amd — MI355X-native hot path of idthanm/env_build.

Host code is Python calling hand-written HIP kernels (gfx950) through the C-ABI of
include/envbuild.h.  There is no CPU fallback: using any compute entry point without
env_build_amd/lib/libenvbuild_hip.so and a visible MI355X raises."""
__version__ = '0.1.0'
