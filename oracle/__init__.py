"""CPU oracle for the profile-HMM Viterbi path.  TEST INFRASTRUCTURE -- see oracle/oracle.py."""
