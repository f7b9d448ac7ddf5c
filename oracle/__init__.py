"""TEST INFRASTRUCTURE ONLY.

CPU restatement (plain PyTorch fp32 / numpy / C) of the reference's hot path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything from this package; the product package never does.
"""
