"""lewton_amd -- MI355X-native Vorbis audio-packet decode path behind lewton's `audio` API surface.

Submodules import the HIP library lazily; `import lewton_amd.streamgen` works without it.
"""
__all__ = ["audio", "header", "batch", "streamgen", "build"]
