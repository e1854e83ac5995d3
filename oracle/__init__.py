"""CPU oracle (test infrastructure only): see oracle/rt_oracle.cpp."""
