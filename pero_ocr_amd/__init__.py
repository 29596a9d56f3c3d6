"""pero_ocr_amd - MI355X-native implementation of pero-ocr's batched text-line
recognition hot path (PytorchEngineLineOCR.process_lines / run_ocr).

Layout:
  csrc/            hand-written HIP kernels (gfx950) + the C ABI (include/pocr.h)
  _native.py       ctypes binding of libpocr_hip.so
  ocr_engine/      host-side mirror of the reference's engine interface
  document_ocr/    PageOCR counterpart (the caller of the hot path)
  netspec.py       model spec, weight-blob format, seeded weight generator
  synth.py         seeded synthetic line crops / charsets
  sharding.py      multi-GPU chunk sharding + RCCL all-gather of decoded labels
"""
__version__ = "0.1.0"
