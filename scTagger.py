#!/usr/bin/env python3
"""Drop-in entry point: `scTagger.py match_trie -lr ... -sr ... [-mr 2] [-bl 16] [-t 16] [-o out.tsv.gz]`."""
from sctagger_amd.cli import main

if __name__ == "__main__":
    main()
