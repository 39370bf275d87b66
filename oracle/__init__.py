"""ORACLE package — test infrastructure only (see mls_oracle.py)."""
