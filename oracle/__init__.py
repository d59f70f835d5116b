"""CPU oracle package -- test infrastructure only (see auction_oracle.c)."""
