from maskbit_amd.bert import Bert, LFQBert  # noqa: F401
