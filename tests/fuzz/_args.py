"""Command lines of the fuzz / stress scripts: NAMED options only, with hard caps.

Round 3 lost two GPU boxes to `fuzz_oneshot.py 100000 8106` -- a seed typed where that script (alone among the
fuzzers) took a thread count as its second positional argument: eight thousand threads, a context each.  So: no
positional arguments anywhere under tests/fuzz/ and tools/ (argparse rejects them), every count has a ceiling that
the script cannot be talked out of, and nothing here starts more than MAX_THREADS threads or MAX_PROCS processes."""
import argparse

MAX_THREADS = 8
MAX_PROCS = 4
MAX_CASES = 1_000_000
MAX_SECONDS = 3600.0


def _bounded(kind, lo, hi):
    def conv(text):
        v = kind(text)
        if not lo <= v <= hi:
            raise argparse.ArgumentTypeError("must be in %s..%s, got %s" % (lo, hi, text))
        return v
    return conv


def parser(description, cases=None, seed=None, threads=None, seconds=None):
    """An ArgumentParser with the options a script asks for (pass the default to get the option)."""
    p = argparse.ArgumentParser(description=description, allow_abbrev=False)
    if cases is not None:
        p.add_argument("--cases", type=_bounded(int, 1, MAX_CASES), default=cases, help="cases / calls to run (default %d)" % cases)
    if seed is not None:
        p.add_argument("--seed", type=_bounded(int, 0, 2**31 - 1), default=seed, help="random seed (default %d)" % seed)
    if threads is not None:
        p.add_argument("--threads", type=_bounded(int, 1, MAX_THREADS), default=threads,
                       help="host threads, at most %d (default %d)" % (MAX_THREADS, threads))
    if seconds is not None:
        p.add_argument("--seconds", type=_bounded(float, 0.1, MAX_SECONDS), default=seconds, help="run time (default %g)" % seconds)
    return p
