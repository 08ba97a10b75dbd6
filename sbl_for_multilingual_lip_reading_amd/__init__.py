"""MI355X-native (gfx950) implementation of the SBL lip-reading forward/backward hot path.

Drop-in surface: `config` and `transformer.{encoder,decoder,transformer,loss,optimizer,...}` mirror the
reference's SBL_Multilingual_Lip_reading package; put this directory on sys.path to import them under the
reference's own top-level names.  All compute runs in libsbl_hip.so (include/sbl_hip.h).
"""
__all__ = ["config", "detfill", "ops", "transformer"]
