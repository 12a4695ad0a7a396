"""CPU oracle (test infrastructure only -- see ddsp_oracle.py).  Never imported by the product."""
