"""CPU oracle package -- test infrastructure only (see oracle/qg_oracle.c)."""
