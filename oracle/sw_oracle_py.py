"""Pure-Python transliteration of the reference's serial aligner -- TEST INFRASTRUCTURE ONLY.

An independent second restatement (small inputs only) used to cross-check
oracle/sw_oracle.c.  "parity unpinned": the reference has no fixtures and cannot
be run here (no JVM); see oracle/sw_oracle.c.

Follows /root/reference/src/sw/SmithWaterman.java statement by statement:
  OptAlignments.call :62-92, ScoreMatrix.call :129-190, GetCellScore.call :217-252,
  InsDelScore.call :277-280, AlignmentScore.call :309-318, GetAlignment.call :354-436;
and /root/reference/src/sw/Distribution.java MapRef.call :403-436.
"""

ALIGN_SCORES = (5, -3, -4)          # Distribution.java:36
ALIGN_TYPES = ("a", "i", "d", "-")  # Distribution.java:37
GAP_CHAR = "_"                      # SmithWaterman.java:356


def _i32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


def _upper(c):
    # Character.toUpperCase for ISO-8859-1 input (the oracle's stated domain)
    o = ord(c)
    if 0x61 <= o <= 0x7A or (0xE0 <= o <= 0xFE and o != 0xF7):
        return chr(o - 32)
    return c


def ins_del_score(cell_score, gap_score):            # :277-280
    return _i32(cell_score + gap_score)


def alignment_score(nw_score, bases, align_scores):   # :309-318
    if _upper(bases[0]) == _upper(bases[1]):
        return _i32(nw_score + align_scores[0])
    return _i32(nw_score + align_scores[1])


def get_cell_score(cell_scores, bases, align_scores, align_types, strict=False):  # :217-252
    mx = 0
    alignment = align_types[3]
    ge = (lambda a, b: a > b) if strict else (lambda a, b: a >= b)   # DistributedSW.java:310,318,326
    tmp = ins_del_score(cell_scores[2], align_scores[2])             # deletion (W)
    if ge(tmp, mx):
        mx, alignment = tmp, align_types[2]
    tmp = ins_del_score(cell_scores[1], align_scores[2])             # insertion (N)
    if ge(tmp, mx):
        mx, alignment = tmp, align_types[1]
    tmp = alignment_score(cell_scores[0], bases, align_scores[:2])   # alignment (NW)
    if ge(tmp, mx):
        mx, alignment = tmp, align_types[0]
    return mx, alignment


def score_matrix(ref_seq, in_seq, align_scores, align_types, strict=False):  # :129-190
    m, n = len(in_seq), len(ref_seq)
    scores = [[0] * (n + 1) for _ in range(m + 1)]
    aligns = [[align_types[3]] * (n + 1) for _ in range(m + 1)]
    max_cells, max_score = [], 0
    if not strict:
        order = ((i, j) for i in range(1, m + 1) for j in range(1, n + 1))
    else:  # DistributedSW.java:192-245: anti-diagonals, ascending j within (CellResultComp)
        order = ((d - j, j) for d in range(2, m + n + 1)
                 for j in range(max(1, d - m), min(n, d - 1) + 1))
    for i, j in order:
        cells = (scores[i - 1][j - 1], scores[i - 1][j], scores[i][j - 1])
        bases = (ref_seq[j - 1], in_seq[i - 1])
        score, t = get_cell_score(cells, bases, align_scores, align_types, strict)
        scores[i][j], aligns[i][j] = score, t
        if score > max_score:
            max_cells = [(i, j)]
            max_score = score
        elif score == max_score:
            max_cells.append((i, j))
    return max_score, max_cells, scores, aligns


def get_alignment(cell, ref_seq, in_seq, align_types, scores, aligns):  # :354-436
    stack = []
    i, j = cell
    score = scores[i][j]
    beginning = 0
    while score > 0:
        beginning = j
        align = aligns[i][j]
        if align == align_types[0]:
            stack.append((ref_seq[j - 1], in_seq[i - 1])); i -= 1; j -= 1
        elif align == align_types[1]:
            stack.append((GAP_CHAR, in_seq[i - 1])); i -= 1
        else:
            stack.append((ref_seq[j - 1], GAP_CHAR)); j -= 1
        score = scores[i][j]
    ref, inn = [], []
    while stack:
        b = stack.pop()
        ref.append(b[0]); inn.append(b[1])
    return beginning, ("".join(ref), "".join(inn))


def opt_alignments(seqs, align_scores=ALIGN_SCORES, align_types=ALIGN_TYPES, strict=False):  # :62-92
    ref_seq, in_seq = seqs
    max_score, max_cells, scores, aligns = score_matrix(ref_seq, in_seq, align_scores, align_types, strict)
    opt = [get_alignment(c, ref_seq, in_seq, align_types, scores, aligns) for c in max_cells]
    if strict:
        opt.sort(key=lambda t: t[0])         # DistributedSW.java:480 (stable)
    return max_score, opt


def map_ref(ref, reads, align_scores=ALIGN_SCORES, align_types=ALIGN_TYPES, strict=False):
    """Distribution.java:403-436.  ref = (metadata, sequence)."""
    total, match_sites = 0, []
    for read in reads:
        s, al = opt_alignments((ref[1], read), align_scores, align_types, strict)
        total = _i32(total + s)
        match_sites.extend(al)
    match_sites.sort(key=lambda t: t[0])     # Collections.sort is stable; MatchSiteComp :691-694
    return total, (ref, match_sites)
