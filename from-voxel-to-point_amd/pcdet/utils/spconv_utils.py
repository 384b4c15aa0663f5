"""pcdet.utils.spconv_utils: the reference module only re-exports spconv; kept for import compatibility."""
from pcdet.ops import spconv  # noqa: F401
