"""Import-path shim: ``modeling.bert``, ``modeling.conv_vqgan`` and ``modeling.modules`` resolve to the
MI355X-native implementations in ``maskbit_amd`` so that the reference's drivers
(scripts/eval_maskbit.py:11-13, demo_utils.py:14-16) import unchanged."""
