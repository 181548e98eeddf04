"""maskbit_amd: MI355X-native MaskBit sampling engine (hand-written gfx950 HIP kernels behind the
reference's LFQBert / ConvVQModel / sample() call surface)."""
from .base_model import BaseModel
from .bert import Bert, LFQBert
from .conv_vqgan import ConvVQModel
from .factorization import combine_factorized_tokens, split_factorized_tokens
from .masking import get_masking_ratio
from .sampling import sample
from .harness import eval_labels, generate_uint8, mask_token_for, to_evaluator_uint8

__all__ = ["BaseModel", "Bert", "LFQBert", "ConvVQModel", "sample", "get_masking_ratio",
           "combine_factorized_tokens", "split_factorized_tokens", "eval_labels", "generate_uint8", "mask_token_for", "to_evaluator_uint8"]
