"""TEST INFRASTRUCTURE ONLY.

`oracle/` holds the CPU restatement of the reference algorithm for the VividMed training step
(`oracle.vividmed`), the shims that let the *reference itself* be imported in the development container
(`oracle.ref_shims`), and the script that generates the committed golden fixtures (`oracle.make_golden`).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything from here, and
only as the checker — never as the thing measured or shipped. The product path (mmmm_amd/) has no CPU
fallback and never imports this package.
"""
