"""Resolves the sibling modules whether this package is imported as
`sbl_for_multilingual_lip_reading_amd.transformer` or, drop-in style, as the top-level `transformer`
(with sbl_for_multilingual_lip_reading_amd/ on sys.path, like the reference's own layout)."""
try:
    from .. import _lib, config, ops            # package mode
except (ImportError, ValueError):
    import _lib                                  # drop-in mode
    import config
    import ops

__all__ = ["_lib", "config", "ops"]
