"""CPU restatement of the reference's SAM-NeRF hot path (test infrastructure only: imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg -- never by the product path)."""
