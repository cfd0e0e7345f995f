from .ms_deform_attn import MSDeformAttn  # noqa: F401
from .encoder_layer import DeformableTransformerEncoderLayer  # noqa: F401
