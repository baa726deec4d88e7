/*
 * swmi_io.h -- C ABI of the native sequence-file reader (part of libswmi.so): SURVEY.md section 8(f)-1.
 *
 * Replaces, with the same quirks, the java.util.Scanner line loops of
 *   InOutOps.GetReads.call    src/sw/InOutOps.java:60-88    (a "reads" file)
 *   InOutOps.GetRefSeqs.call  src/sw/InOutOps.java:115-168  (a reference FASTA file)
 *   InOutOps.IsMetadata.call  src/sw/InOutOps.java:405-411
 * The file is mmap'ed and split with memchr; the result is a packed byte blob + offset table, which is
 * exactly what swmi_batch_upload / swmi_align_batch take, so a file goes disk -> HBM without per-line objects.
 *
 * Line model = Scanner.nextLine on ASCII / ISO-8859-1 text: terminators \n, \r\n, \r; a last line without
 * terminator counts; a trailing terminator does not open an empty line.  (Scanner's U+0085/U+2028/U+2029
 * terminators cannot occur in single-byte DNA files read as UTF-8 and are not handled.)
 */
#ifndef SWMI_IO_H
#define SWMI_IO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct swmi_seqset swmi_seqset;

/* GetReads: line 1, trimmed, is skipped iff it starts with `delimiter`; EVERY following line, trimmed
 * (String.trim: leading/trailing chars <= ' '), is one read -- blank lines (empty reads) and later '>' lines
 * included (InOutOps.java:69-76).  An empty file is SWMI_ERR_INVALID (the reference throws NoSuchElementException). */
int swmi_io_read_reads(const char *path, const char *delimiter, swmi_seqset **out);

/* GetRefSeqs: a line starting with `delimiter` opens a record {metadata = the line, sequence}; every other line
 * is appended UNTRIMMED to the current sequence (InOutOps.java:127-150).  A file that is empty or does not
 * start with a metadata line is SWMI_ERR_INVALID (the reference dies with a NullPointerException at :148/:153). */
int swmi_io_read_refs(const char *path, const char *delimiter, swmi_seqset **out);

uint32_t        swmi_seqset_count(const swmi_seqset *s);
const uint8_t  *swmi_seqset_bytes(const swmi_seqset *s);     /* all sequences back to back            */
const uint64_t *swmi_seqset_offsets(const swmi_seqset *s);   /* count+1 entries, offsets[0] == 0      */
const char     *swmi_seqset_metadata(const swmi_seqset *s, uint32_t k);   /* refs only; "" for reads  */
void            swmi_seqset_free(swmi_seqset *s);

#ifdef __cplusplus
}
#endif
#endif
