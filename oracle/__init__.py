"""CPU checker for the HIP path (test infrastructure; see ctr_oracle.py)."""
