from maskbit_amd.base_model import BaseModel  # noqa: F401
from maskbit_amd.factorization import combine_factorized_tokens, split_factorized_tokens  # noqa: F401
from maskbit_amd.masking import get_masking_ratio  # noqa: F401
from maskbit_amd.sampling import sample  # noqa: F401
