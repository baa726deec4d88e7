"""Pure-Python restatement of the reference's file I/O and control driver -- TEST INFRASTRUCTURE ONLY.

"parity unpinned": the reference ships no fixtures and cannot be run here (no JVM).  Follows
  InOutOps.GetReads.call     /root/reference/src/sw/InOutOps.java:60-88
  InOutOps.GetRefSeqs.call   :115-168        InOutOps.IsMetadata.call :405-411
  InOutOps.GetOutputStr.call :244-288
  Distribution.NoDistribution.call  /root/reference/src/sw/Distribution.java:482-634 (OptSeqsComp :647-666)
with java.util.Scanner.nextLine / hasNextLine restated for ASCII text (terminators \\r\\n, \\n, \\r).
"""
import os
import re

from oracle import sw_oracle as orc

NEWLINE = os.linesep
TAB = "\t"


def _scanner_lines(path):
    data = open(path, "rb").read().decode("latin-1")
    lines = re.split(r"\r\n|\n|\r", data)
    if lines and lines[-1] == "":      # a trailing terminator does not open another line
        lines.pop()
    return lines


def _java_trim(s):
    b, e = 0, len(s)
    while b < e and s[b] <= " ":
        b += 1
    while e > b and s[e - 1] <= " ":
        e -= 1
    return s[b:e]


def is_metadata(line, delimiter):
    return len(line) >= len(delimiter) and line[:len(delimiter)] == delimiter


def get_reads(path, delimiter):
    lines = _scanner_lines(path)
    if not lines:
        raise ValueError("NoSuchElementException")          # scanner.nextLine() on an empty file, :69
    reads = []
    first = _java_trim(lines[0])
    if not is_metadata(first, delimiter):
        reads.append(first)
    for ln in lines[1:]:
        reads.append(_java_trim(ln))
    return reads


def get_ref_seqs(path, delimiter):
    seqs, ref, seq = [], None, None
    for line in _scanner_lines(path):
        if is_metadata(line, delimiter):
            if ref is not None:
                ref[1] = "".join(seq)
                seqs.append(ref)
            ref, seq = [line, None], []
        else:
            if seq is None:
                raise ValueError("NullPointerException")    # :148
            seq.append(line)
    if ref is None:
        raise ValueError("NullPointerException")            # :153
    ref[1] = "".join(seq)
    seqs.append(ref)
    return seqs


def get_output_str(reads, nums, max_score, exec_time, opt):
    s = []
    s.append("Execution Time = %s ms%s" % (exec_time, NEWLINE))
    s.append(NEWLINE)
    s.append("# Reference Sequences = %s%s" % (nums[0], NEWLINE))
    s.append("# Reads = %s%s" % (nums[1], NEWLINE))
    s.append(NEWLINE)
    s.append("Input:" + NEWLINE)
    for r in reads:
        s.append(r + NEWLINE)
    s.append(NEWLINE)
    s.append("Maximum alignment score = %s" % max_score)
    s.append(NEWLINE)
    for ref, sites in opt:
        s.append("Reference:" + NEWLINE)
        s.append(ref[0] + NEWLINE)
        s.append(ref[1] + NEWLINE)
        s.append(NEWLINE)
        for begin, aligned in sites:
            s.append(TAB + "Index = %s%s" % (begin, NEWLINE))
            s.append(TAB + aligned[0] + NEWLINE)
            s.append(TAB + aligned[1] + NEWLINE)
            s.append(NEWLINE)
    return "".join(s)


def _files_sorted(root):
    for name in sorted(os.listdir(root)):
        p = os.path.join(root, name)
        if os.path.isdir(p):
            for x in _files_sorted(p):
                yield x
        else:
            yield p


def no_distribution(ref_dir, in_dir, delimiter, out_dir, out_name="result", out_ext=".txt",
                    scores=(5, -3, -4), types=b"aid-"):
    """The control driver; returns the list of result-file texts (also written to out_dir)."""
    texts = []
    for input_num, in_file in enumerate(_files_sorted(in_dir), 1):
        reads = get_reads(in_file, delimiter)
        num_refs, mx, opt = 0, 0, []
        for ref_file in _files_sorted(ref_dir):
            ref_seqs = get_ref_seqs(ref_file, delimiter)
            num_refs += len(ref_seqs)
            for ref in ref_seqs:
                total, (_, sites) = orc.map_ref(ref, reads, scores, types)
                if total > mx:
                    mx, opt = total, [(ref, sites)]
                elif total == mx:
                    opt.append((ref, sites))
        opt.sort(key=lambda t: t[0][0])
        text = get_output_str(reads, (num_refs, len(reads)), mx, 0, opt)
        with open("%s/%s%d%s" % (out_dir, out_name, input_num, out_ext), "w", newline="", encoding="latin-1") as f:
            f.write(text)
        texts.append(text)
    return texts
