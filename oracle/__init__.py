"""CPU oracle of the MIDAS SNP pileup path -- test infrastructure only (see pileup_oracle.py)."""
