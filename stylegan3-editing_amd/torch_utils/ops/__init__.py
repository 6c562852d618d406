"""Operator API of the MI355X build: same module and function names as the reference's torch_utils/ops."""
