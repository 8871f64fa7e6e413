"""ORACLE -- test infrastructure, not product code (see oracle/kernels.py)."""
