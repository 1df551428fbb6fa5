"""TEST INFRASTRUCTURE ONLY — CPU oracle for the EVer hot path.

`oracle/` restates the reference's algorithm (Z-Zheng/ever 0.5.6) for the path
ResNet encoder -> FPN / FS-Relation / decoder -> head -> pixel loss in stock PyTorch fp32 on the CPU.
The reference's arithmetic lives in PyTorch's ATen CPU kernels (the reference is pure Python and
pins no torch version); this package composes the same ATen ops in the same order.

Pinning: `oracle/gen_golden.py` imports the real reference from /root/reference (build container
only), checks this restatement against it bit-for-bit on the same weights/inputs and writes the
golden vectors under tests/golden/.  tests/test_oracle_golden.py re-checks the restatement against
those committed vectors wherever the tests run.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
ever_amd (the product) never does.
"""
