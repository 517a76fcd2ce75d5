"""peppa_amd: MI355X-native hot path of gchrupala/peppa's PeppaPig training step.

`peppa_amd.models / loss / metrics / triplet / optimization / util` mirror the reference's
`pig.*` API; the compute runs in hand-written HIP kernels for gfx950 behind the
C ABI of include/peppa_hip.h (peppa_amd/libpeppa_hip.so).  There is no CPU fallback.
"""
__version__ = "0.1.0"
