from maskbit_amd.conv_vqgan import ConvVQModel  # noqa: F401
