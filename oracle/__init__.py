"""Test infrastructure only: CPU restatement of the reference algorithm (see crowdstep_oracle.c)."""
