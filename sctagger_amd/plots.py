"""The optional -p plots (host-side, matplotlib; imported only when a plot file is requested).

match_trie: the reference advertises `-p` (README.md:121,134) but its plotting function is never called
(scTagger.py:719-755); the figure it would draw — number of long reads per minimum edit distance — is drawn here.
extract_lr_bc: reads per adapter edit distance with the cumulative count and percentage (scTagger.py:323-359)."""
import numpy as np


def _pyplot():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    return plt


def plot_match_trie(min_e, plotfile, no_hit=0xFF):
    plt = _pyplot()
    vals = np.where(min_e == no_hit, -1, min_e.astype(np.int64))
    labels, counts = np.unique(vals, return_counts=True)
    fig = plt.figure()
    ax = fig.add_subplot(111)
    ax.bar([str(v) for v in labels], counts, color="red", width=0.2)
    ax.set_ylabel("Number of long-reads")
    ax.set_xlabel("Edit distance")
    fig.savefig(plotfile)
    plt.close(fig)


def plot_extract_lr_bc(dist, plotfile):
    plt = _pyplot()
    import matplotlib.ticker as mtick
    labels, counts = np.unique(dist, return_counts=True)
    order = np.argsort(np.where(labels < 0, np.iinfo(np.int64).max, labels), kind="stable")   # 'NA' (-1) last
    labels, counts = labels[order], counts[order]
    names = ["NA" if v < 0 else str(v) for v in labels]
    cum = np.cumsum(counts)
    x = np.arange(labels.size)
    fig = plt.figure(figsize=(10, 5))
    ax = fig.add_subplot(111)
    ax2 = ax.twinx()
    ax.bar(x - 0.1, counts, width=0.2, color="red")
    ax.bar(x + 0.1, cum, width=0.2, color="blue")
    ax2.plot(x + 0.1, cum / max(1, dist.size) * 100, color="blue", alpha=0.0)
    ax.set_xticks(x)
    ax.set_xticklabels(names)
    ax.set_ylabel("Number of Long-reads")
    ax.set_xlabel("Edit distance")
    ax2.yaxis.set_major_formatter(mtick.PercentFormatter())
    ax2.set_ylabel("Percentage of Long-reads")
    fig.savefig(plotfile)
    plt.close(fig)
