// swmi_io.cpp -- native reader for the reference's two input formats (include/swmi_io.h).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/swmi.h"
#include "../../include/swmi_io.h"

struct swmi_seqset {
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> off{0};
    std::vector<std::string> meta;
};

extern "C" const char *swmi_last_error(void);
int swmi_io_fail(int code, const std::string &msg);   // defined in swmi_api.cpp (thread-local error string)

namespace {

struct Mapped {
    const uint8_t *p = nullptr;
    size_t n = 0;
    int fd = -1;
    ~Mapped() {
        if (p && n) munmap((void *)p, n);
        if (fd >= 0) close(fd);
    }
};

int map_file(const char *path, Mapped &m) {
    m.fd = open(path, O_RDONLY);
    if (m.fd < 0) return swmi_io_fail(SWMI_ERR_INVALID, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(m.fd, &st) != 0) return swmi_io_fail(SWMI_ERR_INVALID, std::string("cannot stat ") + path);
    m.n = (size_t)st.st_size;
    if (m.n == 0) return SWMI_OK;
    void *q = mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE, m.fd, 0);
    if (q == MAP_FAILED) { m.n = 0; return swmi_io_fail(SWMI_ERR_NOMEM, std::string("cannot mmap ") + path); }
    madvise(q, m.n, MADV_SEQUENTIAL);
    m.p = (const uint8_t *)q;
    return SWMI_OK;
}

// Scanner.nextLine: returns [b, e) of the next line and advances pos past its terminator
bool next_line(const uint8_t *p, size_t n, size_t &pos, size_t &b, size_t &e) {
    if (pos >= n) return false;
    b = pos;
    const uint8_t *nl = (const uint8_t *)memchr(p + pos, '\n', n - pos);
    const uint8_t *cr = (const uint8_t *)memchr(p + pos, '\r', (nl ? (size_t)(nl - p) : n) - pos);
    if (cr) {                                   // "\r\n" or a lone "\r"
        e = (size_t)(cr - p);
        pos = e + 1;
        if (pos < n && p[pos] == '\n') pos++;
    } else if (nl) {
        e = (size_t)(nl - p);
        pos = e + 1;
    } else {
        e = n;
        pos = n;
    }
    return true;
}

// InOutOps.IsMetadata (InOutOps.java:405-411)
bool is_metadata(const uint8_t *b, size_t len, const char *delim, size_t dlen) {
    return len >= dlen && memcmp(b, delim, dlen) == 0;
}

void trim(const uint8_t *p, size_t &b, size_t &e) {        // String.trim: strip chars <= ' '
    while (b < e && p[b] <= ' ') b++;
    while (e > b && p[e - 1] <= ' ') e--;
}

}  // namespace

extern "C" int swmi_io_read_reads(const char *path, const char *delimiter, swmi_seqset **out) {
    if (!path || !delimiter || !out) return swmi_io_fail(SWMI_ERR_INVALID, "null argument");
    *out = nullptr;
    Mapped m;
    int rc = map_file(path, m);
    if (rc) return rc;
    std::unique_ptr<swmi_seqset> s(new swmi_seqset);
    size_t pos = 0, b, e;
    const size_t dlen = strlen(delimiter);
    if (!next_line(m.p, m.n, pos, b, e))
        return swmi_io_fail(SWMI_ERR_INVALID, std::string("reads file is empty: ") + path);   // NoSuchElementException at :69
    trim(m.p, b, e);
    s->bytes.reserve(m.n);
    if (!is_metadata(m.p + b, e - b, delimiter, dlen)) {                                        // :71-72
        s->bytes.insert(s->bytes.end(), m.p + b, m.p + e);
        s->off.push_back(s->bytes.size());
    }
    while (next_line(m.p, m.n, pos, b, e)) {                                                    // :75-76
        trim(m.p, b, e);
        s->bytes.insert(s->bytes.end(), m.p + b, m.p + e);
        s->off.push_back(s->bytes.size());
    }
    s->meta.assign(s->off.size() - 1, std::string());
    *out = s.release();
    return SWMI_OK;
}

extern "C" int swmi_io_read_refs(const char *path, const char *delimiter, swmi_seqset **out) {
    if (!path || !delimiter || !out) return swmi_io_fail(SWMI_ERR_INVALID, "null argument");
    *out = nullptr;
    Mapped m;
    int rc = map_file(path, m);
    if (rc) return rc;
    std::unique_ptr<swmi_seqset> s(new swmi_seqset);
    s->bytes.reserve(m.n);
    size_t pos = 0, b, e;
    const size_t dlen = strlen(delimiter);
    bool open_rec = false;
    while (next_line(m.p, m.n, pos, b, e)) {
        if (is_metadata(m.p + b, e - b, delimiter, dlen)) {                                     // :131-145
            if (open_rec) s->off.push_back(s->bytes.size());
            s->meta.emplace_back((const char *)m.p + b, e - b);
            open_rec = true;
        } else {
            if (!open_rec)                                                                      // seq is null at :148
                return swmi_io_fail(SWMI_ERR_INVALID, std::string("reference file does not start with a metadata line: ") + path);
            s->bytes.insert(s->bytes.end(), m.p + b, m.p + e);                                  // untrimmed
        }
    }
    if (!open_rec) return swmi_io_fail(SWMI_ERR_INVALID, std::string("reference file has no record: ") + path);   // ref is null at :153
    s->off.push_back(s->bytes.size());
    *out = s.release();
    return SWMI_OK;
}

// ------------------------------------------------------------------------------------------------
// segment-wise GetRefSeqs for the streaming path (swmi_stream_push_file, swmi_api.cpp): the same line rules, applied to
// one slice of the mapped file at a time so that several host threads parse while the GPU aligns earlier slices.
// ------------------------------------------------------------------------------------------------
#include "swmi_io_internal.h"

int swmi_io_map(const char *path, const uint8_t **p, size_t *n, int *fd) {
    Mapped m;
    int rc = map_file(path, m);
    if (rc) return rc;
    *p = m.p; *n = m.n; *fd = m.fd;
    m.p = nullptr; m.n = 0; m.fd = -1;          // ownership moves to the caller
    return SWMI_OK;
}

void swmi_io_unmap(const uint8_t *p, size_t n, int fd) {
    if (p && n) munmap((void *)p, n);
    if (fd >= 0) close(fd);
}

// first position >= from that starts a metadata line (a line beginning with the delimiter), or n
size_t swmi_io_next_record(const uint8_t *p, size_t n, size_t from, const char *delim) {
    const size_t dlen = strlen(delim);
    size_t pos = from;
    if (pos > 0 && pos < n) {
        // move to the start of the next line unless `from` already is one
        const uint8_t prev = p[pos - 1];
        const bool at_start = prev == '\n' || (prev == '\r' && p[pos] != '\n');
        if (!at_start) {
            size_t b, e;
            next_line(p, n, pos, b, e);
        }
    }
    while (pos < n) {
        size_t b, e, q = pos;
        if (!next_line(p, n, q, b, e)) break;
        if (is_metadata(p + b, e - b, delim, dlen)) return b;
        pos = q;
    }
    return n;
}

// parses the records of [from, to) -- `from` must be a metadata line, `to` a record start or n -- appending the sequence
// bytes to dst (capacity >= to - from) exactly as GetRefSeqs does (every non-metadata line untrimmed, InOutOps.java:127-150)
int swmi_io_parse_segment(const uint8_t *p, size_t from, size_t to, const char *delim, uint8_t *dst,
                          std::vector<uint64_t> &off, std::vector<swmi_io_recpos> &recs) {
    const size_t dlen = strlen(delim);
    size_t pos = from, b, e;
    uint64_t at = 0;
    off.clear(); recs.clear();
    off.push_back(0);
    bool open_rec = false;
    while (pos < to && next_line(p, to, pos, b, e)) {
        if (is_metadata(p + b, e - b, delim, dlen)) {
            if (open_rec) { off.push_back(at); recs.back().seq_end = b; }
            swmi_io_recpos r;
            r.meta_pos = b; r.meta_len = e - b; r.seq_pos = pos; r.seq_end = to;
            recs.push_back(r);
            open_rec = true;
        } else {
            if (!open_rec) return swmi_io_fail(SWMI_ERR_INVALID, "reference file does not start with a metadata line");
            memcpy(dst + at, p + b, e - b);
            at += e - b;
        }
    }
    if (open_rec) off.push_back(at);
    return SWMI_OK;
}

// the sequence of one record again (for the alignment strings of a streamed reference: the stream keeps no copy of the bytes)
void swmi_io_read_record(const uint8_t *p, const swmi_io_recpos &r, std::vector<uint8_t> &out) {
    out.clear();
    size_t pos = r.seq_pos, b, e;
    while (pos < r.seq_end && next_line(p, r.seq_end, pos, b, e)) out.insert(out.end(), p + b, p + e);
}

extern "C" uint32_t swmi_seqset_count(const swmi_seqset *s) { return s ? (uint32_t)(s->off.size() - 1) : 0; }
extern "C" const uint8_t *swmi_seqset_bytes(const swmi_seqset *s) { return s ? s->bytes.data() : nullptr; }
extern "C" const uint64_t *swmi_seqset_offsets(const swmi_seqset *s) { return s ? s->off.data() : nullptr; }
extern "C" const char *swmi_seqset_metadata(const swmi_seqset *s, uint32_t k) {
    return (s && k < s->meta.size()) ? s->meta[k].c_str() : "";
}
extern "C" void swmi_seqset_free(swmi_seqset *s) { delete s; }
