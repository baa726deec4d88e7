// swmi_io_internal.h -- pieces of the native FASTA reader (swmi_io.cpp) the streaming path of swmi_api.cpp uses.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string>
#include <vector>

struct swmi_io_recpos {          // where a record sits in the mapped file
    uint64_t meta_pos, meta_len; // its metadata line
    uint64_t seq_pos, seq_end;   // its sequence lines
};

int    swmi_io_fail(int code, const std::string &msg);
int    swmi_io_map(const char *path, const uint8_t **p, size_t *n, int *fd);
void   swmi_io_unmap(const uint8_t *p, size_t n, int fd);
size_t swmi_io_next_record(const uint8_t *p, size_t n, size_t from, const char *delim);
int    swmi_io_parse_segment(const uint8_t *p, size_t from, size_t to, const char *delim, uint8_t *dst,
                             std::vector<uint64_t> &off, std::vector<swmi_io_recpos> &recs);
void   swmi_io_read_record(const uint8_t *p, const swmi_io_recpos &r, std::vector<uint8_t> &out);
