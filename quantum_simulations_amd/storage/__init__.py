"""Reference-compatible on-disk state format (export / import of HBM-resident states)."""
