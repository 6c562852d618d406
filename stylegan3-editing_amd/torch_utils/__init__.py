"""`torch_utils` namespace of the MI355X build.

Same import surface as the reference's `torch_utils` package (misc, persistence, custom_ops, ops.*) so that
unmodified StyleGAN3 network code -- including the module source embedded in official `.pkl` files -- imports
and runs on the HIP kernels in libsg3hip.so.
"""
