"""CPU parity oracle -- test infrastructure only (see oracle/vt_oracle.c header)."""
