"""CPU oracle for the box-processing hot path.  Test infrastructure only (see odbox_oracle.py)."""
