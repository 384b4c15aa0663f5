"""pcdet.ops.DeformableConvolutionV2PyTorch — DCNv2 / DCNv1 behind the reference's module and class names.

The reference installs three TOP-LEVEL names (`DCN`, `functions`, `modules`; DeformableConvolutionV2PyTorch/setup.py)
and its own files import them absolutely (functions/modulated_deform_conv_func.py:13 `import DCN`).  For third-party
code written against that layout `DCN` is also registered under its top-level name (never overriding an existing one)."""
import sys as _sys

from . import DCN as _DCN

_sys.modules.setdefault("DCN", _DCN)
