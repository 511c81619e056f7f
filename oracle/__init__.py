"""CPU oracle (test infrastructure only -- see ppyolo_oracle.py header)."""
