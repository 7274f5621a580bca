"""TEST INFRASTRUCTURE ONLY: CPU oracles for the LBL hot path (see lbl_oracle.py header)."""
