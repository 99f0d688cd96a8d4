/* samgen.c -- fast SAM / FASTA text writer for the synthetic workloads (bench.py's end-to-end leg, tools/).
 * Measurement infrastructure, not part of the product: it turns the record arrays the generator holds
 * (the pp_aln_batch SoA plus the columns a SAM line also carries) into the text an aligner would have
 * written, at memcpy speed, so that the 2 x 1.3 GB files of configs[1] take seconds instead of minutes.
 *
 * Line layout (SURVEY.md section 8d recipe): QNAME "r<read>", FLAG, RNAME, POS (1-based), MAPQ 60, CIGAR, RNEXT "=",
 * PNEXT, TLEN, SEQ, QUAL ('I' x len, or "*"), "NM:i:<nm>".  Records with FLAG & 4 are written unaligned
 * (RNAME "*", POS 0, MAPQ 0, CIGAR "*", no NM tag); records with seq_len == 0 get SEQ "*" / QUAL "*"
 * (secondary alignments of an all-hits run).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char *put_u64(char *p, uint64_t v) {
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

static char *put_i64(char *p, int64_t v) {
    if (v < 0) { *p++ = '-'; return put_u64(p, (uint64_t)(-v)); }
    return put_u64(p, (uint64_t)v);
}

static char *put_str(char *p, const char *s) {
    size_t n = strlen(s);
    memcpy(p, s, n);
    return p + n;
}

typedef struct {
    uint64_t n;                 /* records */
    const uint32_t *read;       /* QNAME = "r<read[i]>" */
    const uint32_t *flag;
    const uint32_t *contig;
    const uint32_t *ref_start;  /* 0-based */
    const uint64_t *cig_off;
    const uint32_t *n_cig;
    const uint32_t *cigar;      /* (len << 4) | op, ops "MIDNSHP=X" */
    const uint32_t *pnext;      /* 0-based mate start */
    const int32_t *tlen;
    const uint64_t *seq_off;
    const uint32_t *seq_len;
    const uint8_t *seq;
    const uint32_t *nm;
    int qual;                   /* 1: 'I' x len, 0: "*" */
} samgen_records;

/* names: n_contigs NUL-terminated strings back to back; lens: contig lengths.  Returns 0, or -1 on an I/O error. */
int samgen_write_sam(const char *path, uint32_t n_contigs, const char *names, const uint64_t *lens,
                     const samgen_records *r) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    const size_t BUF = (size_t)8 << 20;
    char *buf = (char *)malloc(BUF + (1 << 16));
    if (!buf) { fclose(f); return -1; }
    setvbuf(f, NULL, _IONBF, 0);
    const char **nm = (const char **)malloc(sizeof(char *) * (n_contigs ? n_contigs : 1));
    const char *q = names;
    char *p = buf;
    int rc = 0;
    for (uint32_t c = 0; c < n_contigs; c++) {
        nm[c] = q;
        q += strlen(q) + 1;
        p = put_str(p, "@SQ\tSN:");
        p = put_str(p, nm[c]);
        p = put_str(p, "\tLN:");
        p = put_u64(p, lens[c]);
        *p++ = '\n';
        if ((size_t)(p - buf) > BUF) { if (fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) rc = -1; p = buf; }
    }
    static const char OPS[] = "MIDNSHP=X";
    for (uint64_t i = 0; i < r->n && !rc; i++) {
        const uint32_t fl = r->flag[i], sl = r->seq_len[i];
        if ((size_t)(p - buf) + 2ull * sl + 512 + 16ull * r->n_cig[i] > BUF + (1 << 16) - 64 || (size_t)(p - buf) > BUF) {
            if (fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) rc = -1;
            p = buf;
            if (2ull * sl + 512 + 16ull * r->n_cig[i] > BUF) { rc = -1; break; }
        }
        *p++ = 'r';
        p = put_u64(p, r->read[i]);
        *p++ = '\t';
        p = put_u64(p, fl);
        *p++ = '\t';
        if (fl & 4u) {
            p = put_str(p, "*\t0\t0\t*\t*\t0\t0\t");
        } else {
            p = put_str(p, nm[r->contig[i]]);
            *p++ = '\t';
            p = put_u64(p, (uint64_t)r->ref_start[i] + 1u);
            p = put_str(p, "\t60\t");
            const uint32_t *cg = r->cigar + r->cig_off[i];
            for (uint32_t k = 0; k < r->n_cig[i]; k++) {
                p = put_u64(p, cg[k] >> 4);
                *p++ = OPS[cg[k] & 15u];
            }
            p = put_str(p, "\t=\t");
            p = put_u64(p, (uint64_t)r->pnext[i] + 1u);
            *p++ = '\t';
            p = put_i64(p, r->tlen[i]);
            *p++ = '\t';
        }
        if (sl) {
            memcpy(p, r->seq + r->seq_off[i], sl);
            p += sl;
            *p++ = '\t';
            if (r->qual) { memset(p, 'I', sl); p += sl; } else *p++ = '*';
        } else {
            p = put_str(p, "*\t*");
        }
        if (!(fl & 4u)) {
            p = put_str(p, "\tNM:i:");
            p = put_u64(p, r->nm[i]);
        }
        *p++ = '\n';
    }
    if (!rc && p > buf && fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) rc = -1;
    if (fclose(f) != 0) rc = -1;
    free(buf);
    free(nm);
    return rc;
}

/* One sequence per line (what the polish output looks like too). */
int samgen_write_fasta(const char *path, uint32_t n_contigs, const char *names, const uint64_t *off, const uint8_t *bases) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    const char *q = names;
    int rc = 0;
    for (uint32_t c = 0; c < n_contigs && !rc; c++) {
        if (fputc('>', f) == EOF || fputs(q, f) == EOF || fputc('\n', f) == EOF) rc = -1;
        q += strlen(q) + 1;
        const uint64_t len = off[c + 1] - off[c];
        if (!rc && fwrite(bases + off[c], 1, len, f) != len) rc = -1;
        if (!rc && fputc('\n', f) == EOF) rc = -1;
    }
    if (fclose(f) != 0) rc = -1;
    return rc;
}
