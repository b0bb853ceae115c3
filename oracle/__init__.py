"""CPU oracle (test infrastructure only -- see oracle/lr_oracle.c).  Never imported by logreg_amd."""
