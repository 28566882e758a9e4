"""CPU oracle for the match_trie path — TEST INFRASTRUCTURE ONLY (see trie_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
