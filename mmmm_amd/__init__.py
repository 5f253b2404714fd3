"""MI355X-native VividMed (function2-llx/MMMM) training step: HIP kernels behind the reference's mmmm.models surface."""
