"""Read correction / assembly polishing end to end: a thin Python face of the library's native driver, cw_run_correction
(consent_amd/csrc/cw_driver.cpp), which is the loop of CONSENT-correction.cpp (runCorrection :62-135, processRead :19-58) and of
CONSENT-polishing.cpp (:107-135, :21-105): host feeders -> device-resident pile extraction -> consensus -> re-assembly, jobs of
piles spread over the GPUs by one queue, FASTA in PAF order.  bin/CONSENT-correction and bin/CONSENT-polishing are the same
function behind the reference's own command line; `python -m consent_amd.pipeline` accepts that command line too.

This module computes nothing: it fills a cw_driver_args, hands the library a file descriptor and reads the FASTA back.
"""
import ctypes as C
import getopt
import os
import sys
import tempfile

# Before the HIP runtime starts (the library is loaded lazily, below): the driver's two workers per GPU have a dozen streams and HIP spreads a
# process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4).  The host process sets it, not the library; a caller's own value wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

from .engine import EngineError, load_library  # noqa: E402


class DriverArgs(C.Structure):
    _fields_ = [
        ("paf_index", C.c_char_p), ("alignment_file", C.c_char_p), ("reads_file", C.c_char_p), ("proof_file", C.c_char_p), ("path", C.c_char_p),
        ("min_support", C.c_uint32), ("max_support", C.c_uint32), ("window_size", C.c_uint32), ("mer_size", C.c_uint32), ("common_kmers", C.c_uint32),
        ("min_anchors", C.c_uint32), ("solid_thresh", C.c_uint32), ("window_overlap", C.c_uint32), ("nb_threads", C.c_uint32), ("max_msa", C.c_uint32),
        ("polishing", C.c_int32), ("devices", C.POINTER(C.c_int32)), ("n_devices", C.c_int32), ("windows_per_batch", C.c_uint32),
    ]


class DriverStats(C.Structure):
    _fields_ = [
        ("n_devices", C.c_uint32), ("piles", C.c_uint64), ("windows", C.c_uint64), ("overlaps", C.c_uint64), ("jobs", C.c_uint64), ("records", C.c_uint64),
        ("bases_out", C.c_uint64), ("ms_index", C.c_double), ("ms_total", C.c_double), ("dev_windows", C.c_uint64 * 16), ("dev_ms_extract", C.c_double * 16),
        ("dev_ms_consensus", C.c_double * 16), ("dev_ms_stitch", C.c_double * 16),
    ]


def run_correction(reads_path, paf_path, out_fd, *, min_support=3, max_support=1000, window_size=500, mer_size=9, common_kmers=8, min_anchors=10, solid_thresh=4,
                   window_overlap=50, max_msa=150, nb_threads=1, proof_path=None, polishing=False, devices=None, windows_per_batch=0):
    """cw_run_correction with runCorrection's parameters (defaults of src/main.cpp:17-26).  FASTA goes to the file descriptor `out_fd`.
    Returns the driver's counters; raises EngineError on any library error (no GPU, malformed input, capacity)."""
    lib = load_library()
    lib.cw_run_correction.argtypes = [C.POINTER(DriverArgs), C.c_int, C.POINTER(DriverStats)]
    dev_arr = (C.c_int32 * len(devices))(*devices) if devices else None
    a = DriverArgs(b"", os.fsencode(paf_path), os.fsencode(reads_path), os.fsencode(proof_path) if proof_path else b"", b"", min_support, max_support, window_size,
                   mer_size, common_kmers, min_anchors, solid_thresh, window_overlap, nb_threads, max_msa, 1 if polishing else 0,
                   C.cast(dev_arr, C.POINTER(C.c_int32)) if devices else None, len(devices) if devices else 0, windows_per_batch)
    st = DriverStats()
    rc = lib.cw_run_correction(C.byref(a), out_fd, C.byref(st))
    if rc != 0:
        raise EngineError(f"cw_run_correction: {lib.cw_strerror(rc).decode()} ({rc})")
    return st


def correct_reads(reads_path, paf_path, out=None, *, min_support=3, max_support=1000, window_size=500, mer_size=9, common_kmers=8, min_anchors=10, solid_thresh=4,
                  window_overlap=50, max_msa=150, do_trim=True, proof_path=None, windows_per_batch=0, device=0, devices=None, polishing=False, on_capacity="raise"):
    """Corrects every read that has a pile in `paf_path`; writes FASTA (">name\\nsequence\\n", upper case = corrected) to `out` (a text
    file object; None = collect only) and returns the list of (name, sequence) in PAF order.  Reads whose corrected sequence is empty
    -- no window, or dropped by the 10 % rule -- are skipped, as CONSENT-correction.cpp:101-103 does.  With `proof_path` (assembly
    polishing, or correction against proof reads) that file is indexed into the same read set and nothing is trimmed or dropped
    (CONSENT-correction.cpp:69-73, CONSENT-polishing.cpp:19); `do_trim=False` asks for the polishing driver's behaviour explicitly.
    `devices`: the GPUs to spread the piles over (an id may repeat: several engines on one GPU); default [device].
    `on_capacity`: a read whose re-assembly (or one of whose windows) exceeded a documented capacity of the engine either stops the run
    ("raise", the default: nothing is silently different from the reference) or is left out and reported on stderr ("skip")."""
    old = os.environ.get("CW_ON_CAPACITY")
    os.environ["CW_ON_CAPACITY"] = "skip" if on_capacity == "skip" else "raise"
    try:
        with tempfile.TemporaryFile() as tmp:
            run_correction(reads_path, paf_path, tmp.fileno(), min_support=min_support, max_support=max_support, window_size=window_size, mer_size=mer_size,
                           common_kmers=common_kmers, min_anchors=min_anchors, solid_thresh=solid_thresh, window_overlap=window_overlap, max_msa=max_msa,
                           proof_path=proof_path, polishing=polishing or not do_trim, devices=list(devices) if devices else [device], windows_per_batch=windows_per_batch)
            tmp.seek(0)
            text = tmp.read().decode()
    finally:
        if old is None:
            os.environ.pop("CW_ON_CAPACITY", None)
        else:
            os.environ["CW_ON_CAPACITY"] = old
    if out is not None:
        out.write(text)
    lines = text.split("\n")
    return [(lines[i][1:], lines[i + 1]) for i in range(0, len(lines) - 1, 2)]


USAGE = "Usage: %s -a overlaps.paf -r reads.fasta [-R reads.fasta] [-s N] [-S N] [-M N] [-l N] [-m N] [-k N] [-c N] [-A N] [-f N] [-j GPUs] (-i, -p accepted and ignored)\n\n"


def main(argv=None, polishing=False):
    """The reference's command line (src/main.cpp:29-76): same getopt string, same defaults; -d -e -w -n fall into its `default` case."""
    argv = sys.argv[1:] if argv is None else argv
    try:
        opts, _ = getopt.getopt(argv, "a:A:d:k:s:S:M:l:f:e:p:c:m:j:w:m:r:R:n:i:")
    except getopt.GetoptError:
        sys.stderr.write(USAGE % "consent_amd.pipeline")
        raise SystemExit(1)
    kw = dict(min_support=3, max_support=1000, max_msa=150, window_size=500, mer_size=9, common_kmers=8, min_anchors=10, solid_thresh=4, window_overlap=50, nb_threads=1)
    names = {"-s": "min_support", "-S": "max_support", "-M": "max_msa", "-l": "window_size", "-k": "mer_size", "-c": "common_kmers", "-A": "min_anchors", "-f": "solid_thresh",
             "-m": "window_overlap", "-j": "nb_threads"}
    paf = reads = ""
    proof = None
    for o, v in opts:
        if o in names:
            kw[names[o]] = int(v)  # atoi
        elif o == "-a":
            paf = v
        elif o == "-r":
            reads = v
        elif o == "-R":
            proof = v
        elif o in ("-i", "-p"):
            pass  # stored and never read by the reference
        else:
            sys.stderr.write(USAGE % "consent_amd.pipeline")
            raise SystemExit(1)
    sys.stdout.flush()
    run_correction(reads, paf, sys.stdout.fileno(), proof_path=proof, polishing=polishing, **kw)


if __name__ == "__main__":
    main(polishing="--polishing" in sys.argv and not sys.argv.remove("--polishing"))
