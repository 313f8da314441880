"""CPU oracle of the plane-sweep / DPV hot path -- test infrastructure only (see ref_cpu.py)."""
