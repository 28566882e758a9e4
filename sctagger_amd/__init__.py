"""sctagger_amd — MI355X-native implementation of scTagger's match_trie stage.

    sctagger_amd.match_trie   host-side mirror of the reference functions for this path
    sctagger_amd.cli          the `scTagger.py match_trie` command line
    sctagger_amd._native      ctypes binding of csrc/libsctagger_hip.so (include/sctagger_hip.h)
    sctagger_amd.synth        seeded synthetic inputs
"""
__version__ = "0.1.0"
