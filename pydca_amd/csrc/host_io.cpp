// Host-side MSA reader with the semantics of PlmDCA::readSequencesFromFile
// (pydca/plmdca/plmdca_numerics.cpp:685-767): every non-empty line not starting with '>'
// is one sequence, only its first L characters are used (upper-cased), residues map to
// 0-based codes with gap = q-1, rows already seen are dropped (first occurrence kept).
// Where the reference throws (unopenable file :743-746, character missing from its table
// :752) this returns an error code instead.
//
// Built for alignments of 10^5 sequences: ONE mmap of the file, a line index from memchr, the rows encoded through a
// 256-entry table (upper-casing folded in) and hashed by a few host threads, first-occurrence de-duplication through an
// open-addressing table of 64-bit row hashes with memcmp on a hash match (file order, so the reference's "first
// occurrence wins" is kept), rows compacted in place.  Config D (25 MB, 50 000 x 500): 0.67 s with getline + a set of
// strings -> a few ms.
#include <cerrno>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>

#include "dca_internal.h"

namespace {

// residue tables of plmdca_numerics.cpp:708-717 (protein) and :722-731 (RNA: ACGU, the three gap
// characters and every other capital letter -- 'T' included, :729 -- are the gap state); lower-case letters
// carry their capital's code (the reference upper-cases every character, :752)
struct CodeTable {
    int8_t protein[256];
    int8_t rna[256];
    int8_t mf_protein[256];      // the Python reader's tables (fasta_reader.py:34-45, :138-149): anything unknown is the gap
    int8_t mf_rna[256];
    CodeTable()
    {
        for (int c = 0; c < 256; ++c) { protein[c] = rna[c] = -1; mf_protein[c] = 20; mf_rna[c] = 4; }
        const char* aa = "ACDEFGHIKLMNPQRSTVWY";
        for (int k = 0; aa[k]; ++k) protein[(unsigned char)aa[k]] = mf_protein[(unsigned char)aa[k]] = (int8_t)k;
        for (const char* p = "-.~BJOUXZ"; *p; ++p) protein[(unsigned char)*p] = 20;
        const char* nt = "ACGU";
        for (int k = 0; nt[k]; ++k) rna[(unsigned char)nt[k]] = mf_rna[(unsigned char)nt[k]] = (int8_t)k;
        for (const char* p = "-~.BDEFHIJKLMNOPQRSTVWXYZ"; *p; ++p) rna[(unsigned char)*p] = 4;
        for (int c = 'a'; c <= 'z'; ++c) {
            protein[c] = protein[c - 32]; rna[c] = rna[c - 32];
            mf_protein[c] = mf_protein[c - 32]; mf_rna[c] = mf_rna[c - 32];
        }
    }
};
const CodeTable kCodes;

// read-only view of a whole file: regular files are mapped; anything else that can be opened and read (a FIFO, /dev/stdin,
// a process substitution -- the reference's std::ifstream / Biopython readers take those too) or a file that cannot be
// mapped is read into a heap buffer, and the same indexer runs over it
struct MappedFile {
    const char* data = nullptr;
    size_t size = 0;
    int fd = -1;
    bool mapped = false;
    std::vector<char> heap;
    bool slurp()
    {
        heap.clear();
        char buf[1 << 16];
        for (;;) {
            const ssize_t got = ::read(fd, buf, sizeof(buf));
            if (got < 0) { if (errno == EINTR) continue; return false; }
            if (got == 0) break;
            heap.insert(heap.end(), buf, buf + got);
        }
        data = heap.data();
        size = heap.size();
        return true;
    }
    bool open(const char* path)
    {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || S_ISDIR(st.st_mode)) { ::close(fd); fd = -1; return false; }
        if (S_ISREG(st.st_mode)) {
            size = (size_t)st.st_size;
            if (size == 0) return true;
            void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
            if (p != MAP_FAILED) {
                madvise(p, size, MADV_SEQUENTIAL);
                data = static_cast<const char*>(p);
                mapped = true;
                return true;
            }
        }
        if (!slurp()) { ::close(fd); fd = -1; return false; }
        return true;
    }
    ~MappedFile()
    {
        if (mapped) munmap(const_cast<char*>(data), size);
        if (fd >= 0) ::close(fd);
    }
};

struct LineRef { const char* p; uint32_t len; };

// every non-empty line that does not start with '>' ('\n' is the only terminator, as std::getline's, :748)
void index_sequence_lines(const MappedFile& f, std::vector<LineRef>& lines)
{
    const char* p = f.data;
    const char* const end = f.data + f.size;
    while (p < end) {
        const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
        const char* e = nl ? nl : end;
        if (e > p && *p != '>') lines.push_back(LineRef{p, (uint32_t)std::min<size_t>((size_t)(e - p), 0xffffffffu)});
        p = e + 1;
    }
}

inline uint64_t mix64(uint64_t h)
{
    h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 32;
    return h;
}
// four independent multiply-xor lanes over the row's 8-byte words (a single chained mix is ~10 cycles per word and was two
// thirds of the reader's time), folded at the end; collisions only cost a memcmp
uint64_t hash_row(const uint8_t* r, int L)
{
    uint64_t h0 = 0x9e3779b97f4a7c15ull ^ (uint64_t)L, h1 = 0xc2b2ae3d27d4eb4full, h2 = 0x165667b19e3779f9ull, h3 = 0x27d4eb2f165667c5ull;
    const uint64_t K = 0xd6e8feb86659fd93ull;
    int s = 0;
    for (; s + 32 <= L; s += 32) {
        uint64_t v[4];
        memcpy(v, r + s, 32);
        h0 = (h0 ^ v[0]) * K; h0 ^= h0 >> 29;
        h1 = (h1 ^ v[1]) * K; h1 ^= h1 >> 29;
        h2 = (h2 ^ v[2]) * K; h2 ^= h2 >> 29;
        h3 = (h3 ^ v[3]) * K; h3 ^= h3 >> 29;
    }
    uint64_t tail[4] = {0, 0, 0, 0};
    memcpy(tail, r + s, (size_t)(L - s));
    h0 = (h0 ^ tail[0]) * K; h1 = (h1 ^ tail[1]) * K; h2 = (h2 ^ tail[2]) * K; h3 = (h3 ^ tail[3]) * K;
    return mix64(h0 ^ mix64(h1 ^ mix64(h2 ^ mix64(h3))));
}

template <typename F> void parallel_rows(size_t n, F&& body)
{
    unsigned nt = std::min<size_t>({(size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)32, (n + 1023) / 1024});
    if (const char* e = getenv("DCA_READER_THREADS")) nt = std::max(1, atoi(e));
    if (nt <= 1) { body((size_t)0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t a = std::min(n, t * per), b = std::min(n, a + per);
        if (a < b) th.emplace_back([&body, a, b] { body(a, b); });
    }
    for (auto& t : th) t.join();
}

// rows[k] (k < n, L bytes each, stride L) -> the first occurrence of every distinct row, in order, compacted to the front
// of `rows`; returns how many.  hashes[k] = hash_row(rows[k]).
size_t dedup_first_occurrence(uint8_t* rows, const uint64_t* hashes, size_t n, int L)
{
    size_t cap = 16;
    while (cap < 2 * n) cap <<= 1;
    std::vector<uint32_t> slot(cap, 0xffffffffu);       // index into the COMPACTED prefix
    std::vector<uint64_t> kept_hash;
    kept_hash.reserve(n);
    size_t kept = 0;
    for (size_t k = 0; k < n; ++k) {
        const uint64_t h = hashes[k];
        const uint8_t* r = rows + k * (size_t)L;
        size_t pos = (size_t)h & (cap - 1);
        bool dup = false;
        while (slot[pos] != 0xffffffffu) {
            const uint32_t j = slot[pos];
            if (kept_hash[j] == h && memcmp(rows + (size_t)j * L, r, (size_t)L) == 0) { dup = true; break; }
            pos = (pos + 1) & (cap - 1);
        }
        if (dup) continue;
        slot[pos] = (uint32_t)kept;
        kept_hash.push_back(h);
        if (kept != k) memmove(rows + kept * (size_t)L, r, (size_t)L);
        ++kept;
    }
    return kept;
}

}  // namespace

int dca_count_msa_lines_impl(const char* path)
{
    MappedFile f;
    if (!path || !f.open(path)) { dca_set_error("Unable to open file %s", path ? path : "(null)"); return DCA_ERR_IO; }
    // same rule as index_sequence_lines, without keeping the index
    int n = 0;
    const char* p = f.data;
    const char* const end = f.data + f.size;
    while (p < end) {
        const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
        const char* e = nl ? nl : end;
        if (e > p && *p != '>') ++n;
        p = e + 1;
    }
    return n;
}

// -> number of unique rows (>= 0) or an error code (< 0).  The rows are encoded straight into `out` (capacity rows of L
// bytes) when the caller has one; otherwise *owned receives a malloc'd block (uninitialised: every byte kept is written).
static int read_msa_core(const char* path, int biomolecule, int L, uint8_t* out, int capacity, uint8_t** owned, int* raw_count)
{
    if (!path || L <= 0 || (biomolecule != DCA_BIOMOLECULE_PROTEIN && biomolecule != DCA_BIOMOLECULE_RNA)) {
        dca_set_error("dca_read_msa: bad arguments");
        return DCA_ERR_ARG;
    }
    auto T0 = std::chrono::steady_clock::now();
    const bool timing = getenv("DCA_READER_TIMING") != nullptr;
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "  reader %-8s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count());
        T0 = t;
    };
    MappedFile f;
    if (!f.open(path)) {
        dca_set_error("Unable to open file %s", path);
        return DCA_ERR_IO;
    }
    lap("mmap");
    std::vector<LineRef> lines;
    lines.reserve(f.size / ((size_t)L + 1) + 16);
    index_sequence_lines(f, lines);
    const size_t n = lines.size();
    if (n > 0x7fffffffu) { dca_set_error("dca_read_msa: too many sequences"); return DCA_ERR_ARG; }
    lap("index");
    const int8_t* table = biomolecule == DCA_BIOMOLECULE_PROTEIN ? kCodes.protein : kCodes.rna;
    uint8_t* R = out;
    uint8_t* mine = nullptr;
    if (!R || (size_t)capacity < n) {          // no buffer, or one sized for the unique rows only: stage here
        mine = static_cast<uint8_t*>(malloc(std::max<size_t>(n * (size_t)L, 1)));
        if (!mine) { dca_set_error("out of host memory"); return DCA_ERR_NOMEM; }
        R = mine;
    }
    std::vector<uint64_t> hashes(n);
    // first line (in file order) the reference would throw on: shorter than L (.at() past the end, :752) or a
    // character its table lacks
    std::atomic<size_t> firstBad(n);
    parallel_rows(n, [&](size_t a, size_t b) {
        for (size_t k = a; k < b; ++k) {
            if (k > firstBad.load(std::memory_order_relaxed)) return;
            const unsigned char* src = reinterpret_cast<const unsigned char*>(lines[k].p);
            uint8_t* dst = R + k * (size_t)L;
            int bad = lines[k].len < (uint32_t)L;
            if (!bad) {
                int acc = 0;
                for (int s = 0; s < L; ++s) { const int8_t c = table[src[s]]; acc |= c; dst[s] = (uint8_t)c; }
                bad = acc < 0;
            }
            if (bad) {
                size_t cur = firstBad.load();
                while (k < cur && !firstBad.compare_exchange_weak(cur, k)) {}
                return;
            }
            hashes[k] = hash_row(dst, L);
        }
    });
    lap("encode");
    if (firstBad.load() < n) {
        const size_t k = firstBad.load();
        if (lines[k].len < (uint32_t)L) {
            dca_set_error("sequence line %zu of %s is shorter than %d", k + 1, path, L);
        } else {
            int s = 0;
            while (table[(unsigned char)lines[k].p[s]] >= 0) ++s;
            dca_set_error("character '%c' of sequence line %zu is not in the residue table", lines[k].p[s], k + 1);
        }
        free(mine);
        return DCA_ERR_RESIDUE;
    }
    const size_t kept = dedup_first_occurrence(R, hashes.data(), n, L);
    lap("dedup");
    if (raw_count) *raw_count = (int)n;
    if (mine && out) {                          // staged because the caller's buffer holds the unique rows only
        if (kept > (size_t)capacity) { free(mine); dca_set_error("dca_read_msa: capacity %d too small", capacity); return DCA_ERR_ARG; }
        memcpy(out, mine, kept * (size_t)L);
        free(mine);
    } else if (mine) {
        if (owned) *owned = mine; else free(mine);
    }
    return (int)kept;
}

int dca_read_msa_owned(const char* path, int biomolecule, int L, uint8_t** rows, int* raw_count)
{
    *rows = nullptr;
    return read_msa_core(path, biomolecule, L, nullptr, 0, rows, raw_count);
}

int dca_read_msa_impl(const char* path, int biomolecule, int L, uint8_t* out, int capacity, int* raw_count)
{
    return read_msa_core(path, biomolecule, L, out, capacity, nullptr, raw_count);
}

// ---------------------------------------------------------------------------------------------
// FASTA reader of the mfDCA path: the semantics of pydca/fasta_reader/fasta_reader.py:81-163 as read through
// Biopython -- records start at '>' lines, a record's sequence is the concatenation of its stripped lines, records
// without residues are dropped, letters are upper-cased, anything outside the alphabet is the gap state (:138-149),
// exact duplicates are dropped keeping the first occurrence (:153).  Codes are 0-based here with gap = q - 1 (the
// Python side adds 1 where the reference's 1-based states are handed out).  Pure-ASCII files only: a byte >= 0x80
// returns DCA_ERR_RESIDUE and the caller falls back to its own text-mode reader.
// dca_fasta_shape: *n_records = records with residues, *L = their common length (DCA_ERR_ARG when they differ).
namespace {
inline bool py_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }

struct FastaPiece { const char* p; uint32_t len; };
struct FastaRecord { uint32_t first, npieces; size_t len; };

int index_fasta(const MappedFile& f, std::vector<FastaPiece>& pieces, std::vector<FastaRecord>& recs)
{
    const char* p = f.data;
    const char* const end = f.data + f.size;
    bool in_record = false;
    FastaRecord cur{0, 0, 0};
    auto close_record = [&] {
        if (!in_record) return;
        if (cur.len > 0) recs.push_back(cur);
        else pieces.resize(cur.first);
    };
    // universal newlines, as Python's text mode: '\n', '\r\n' and a lone '\r' all end a line.  Files without any '\r' (the
    // usual case; one memchr over the file tells) are split with memchr instead of byte by byte
    const bool anyCR = f.size && memchr(f.data, '\r', f.size) != nullptr;
    while (p < end) {
        const char* e;
        if (anyCR) {
            e = p;
            while (e < end && *e != '\n' && *e != '\r') ++e;
        } else {
            e = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
            if (!e) e = end;
        }
        const char* a = p;
        const char* b = e;
        while (a < b && py_space((unsigned char)*a)) ++a;
        while (b > a && py_space((unsigned char)b[-1])) --b;
        if (b > a) {
            if (*a == '>') {
                close_record();
                in_record = true;
                cur = FastaRecord{(uint32_t)pieces.size(), 0, 0};
            } else if (in_record) {
                pieces.push_back(FastaPiece{a, (uint32_t)(b - a)});
                cur.npieces += 1;
                cur.len += (size_t)(b - a);
            }
        }
        p = e + 1;
    }
    close_record();
    return DCA_OK;
}
}  // namespace

static bool has_non_ascii(const MappedFile& f)
{
    uint64_t acc = 0;
    size_t k = 0;
    for (; k + 8 <= f.size; k += 8) { uint64_t v; memcpy(&v, f.data + k, 8); acc |= v; }
    for (; k < f.size; ++k) acc |= (uint64_t)(unsigned char)f.data[k];
    return (acc & 0x8080808080808080ull) != 0;
}

extern "C" int dca_fasta_shape(const char* path, int* n_records, int* L_out)
{
    MappedFile f;
    if (!path || !n_records || !L_out) return DCA_ERR_ARG;
    if (!f.open(path)) { dca_set_error("Unable to open file %s", path); return DCA_ERR_IO; }
    // non-ASCII bytes anywhere: byte lengths are not character counts, the text-mode reader has to take this file
    if (has_non_ascii(f)) { dca_set_error("%s holds non-ASCII bytes", path); return DCA_ERR_RESIDUE; }
    std::vector<FastaPiece> pieces;
    std::vector<FastaRecord> recs;
    index_fasta(f, pieces, recs);
    *n_records = (int)recs.size();
    *L_out = recs.empty() ? 0 : (int)recs[0].len;
    for (const FastaRecord& r : recs)
        if (r.len != recs[0].len) { dca_set_error("Sequences in %s do not all have the same length", path); return DCA_ERR_ARG; }
    return DCA_OK;
}

// One pass: index, encode, de-duplicate.  With `out` (capacity rows of L bytes, L as dca_fasta_shape reported it) the rows
// go there; with `owned` the reader allocates (malloc; dca_host_free) and reports the length it found in *L_io.
// Returns the number of unique rows (>= 0) or an error code.
static int read_fasta_core(const char* path, int biomolecule, int* L_io, uint8_t* out, int capacity, uint8_t** owned, int* raw_count)
{
    if (!path || !L_io || (!out && !owned) || (biomolecule != DCA_BIOMOLECULE_PROTEIN && biomolecule != DCA_BIOMOLECULE_RNA)) {
        dca_set_error("dca_read_fasta: bad arguments");
        return DCA_ERR_ARG;
    }
    MappedFile f;
    if (!f.open(path)) { dca_set_error("Unable to open file %s", path); return DCA_ERR_IO; }
    std::vector<FastaPiece> pieces;
    std::vector<FastaRecord> recs;
    index_fasta(f, pieces, recs);
    const size_t n = recs.size();
    const int L = out ? *L_io : (n ? (int)recs[0].len : 0);
    *L_io = L;
    if (out && n > (size_t)capacity) { dca_set_error("dca_read_fasta: capacity %d too small", capacity); return DCA_ERR_ARG; }
    for (const FastaRecord& r : recs)
        if (r.len != (size_t)L) {
            // byte lengths are not character counts in a file with multi-byte characters: that one is the text-mode reader's
            if (has_non_ascii(f)) { dca_set_error("%s holds non-ASCII bytes", path); return DCA_ERR_RESIDUE; }
            dca_set_error("Sequences in %s do not all have the same length", path);
            return DCA_ERR_ARG;
        }
    if (!out) {
        out = static_cast<uint8_t*>(malloc(std::max<size_t>(n * (size_t)L, 1)));
        if (!out) { dca_set_error("out of host memory"); return DCA_ERR_NOMEM; }
        *owned = out;
    }
    const int8_t* table = biomolecule == DCA_BIOMOLECULE_PROTEIN ? kCodes.mf_protein : kCodes.mf_rna;
    std::vector<uint64_t> hashes(n);
    std::atomic<int> nonAscii(0);
    parallel_rows(n, [&](size_t a, size_t b) {
        for (size_t k = a; k < b; ++k) {
            uint8_t* dst = out + k * (size_t)L;
            int hi = 0;
            size_t o = 0;
            for (uint32_t pc = 0; pc < recs[k].npieces; ++pc) {
                const FastaPiece& P = pieces[recs[k].first + pc];
                const unsigned char* src = reinterpret_cast<const unsigned char*>(P.p);
                for (uint32_t s = 0; s < P.len; ++s) { hi |= src[s]; dst[o + s] = (uint8_t)table[src[s]]; }
                o += P.len;
            }
            if (hi & 0x80) nonAscii.store(1);
            hashes[k] = hash_row(dst, L);
        }
    });
    if (nonAscii.load()) {
        if (owned && *owned) { free(*owned); *owned = nullptr; }
        dca_set_error("%s holds non-ASCII bytes", path);
        return DCA_ERR_RESIDUE;
    }
    const size_t kept = dedup_first_occurrence(out, hashes.data(), n, L);
    if (raw_count) *raw_count = (int)n;
    return (int)kept;
}

extern "C" int dca_read_fasta(const char* path, int biomolecule, int L, uint8_t* out, int capacity, int* raw_count)
{
    if (!out || L <= 0) { dca_set_error("dca_read_fasta: bad arguments"); return DCA_ERR_ARG; }
    return read_fasta_core(path, biomolecule, &L, out, capacity, nullptr, raw_count);
}

extern "C" int dca_read_fasta_alloc(const char* path, int biomolecule, uint8_t** rows, int* L_out, int* raw_count)
{
    if (!rows || !L_out) { dca_set_error("dca_read_fasta_alloc: bad arguments"); return DCA_ERR_ARG; }
    *rows = nullptr;
    *L_out = 0;
    return read_fasta_core(path, biomolecule, L_out, nullptr, 0, rows, raw_count);
}

extern "C" void dca_host_free(void* p) { free(p); }

// ---------------------------------------------------------------------------------------------
// Local pairwise alignment (Smith-Waterman with Gotoh's affine gaps) for the reference-sequence
// back-mapping (SequenceBackmapper.align_pairs_local, sequence_backmapper.py:186-230, which calls
// Bio.pairwise2.align.localds -- biopython 1.74, not part of the reference tree).  Gap of length n
// costs open + (n-1)*extend as in pairwise2 (penalize_extend_when_opening = False).  sub: 26x26
// substitution scores indexed by letter - 'A'.
//
// dca_sw_scores: best local score of `ref` against each of `nseq` sequences packed back to back in
// `seqs` (offsets[k] .. offsets[k+1]); this is the search loop of find_matching_seqs_from_alignment
// (:233-283).  dca_sw_align: one alignment with traceback.  Tie rules (pairwise2 returns a list and
// the reference takes element 0; its ordering is not documented): the end cell is the first maximum
// in row-major order and the traceback prefers diagonal, then a gap in the second sequence, then a
// gap in the reference.
namespace {
struct SwRows {
    std::vector<int> H, E;
};

inline int sub_score(const int* sub, unsigned char a, unsigned char b)
{
    const int ia = a - 'A', ib = b - 'A';
    if (ia < 0 || ia >= 26 || ib < 0 || ib >= 26) return -4;
    return sub[ia * 26 + ib];
}

int sw_score_one(const char* a, int la, const char* b, int lb, const int* sub, int open, int ext, SwRows& w)
{
    // rows over a (reference), columns over b
    w.H.assign(lb + 1, 0);
    w.E.assign(lb + 1, -(1 << 28));      // best score ending in a gap in a (vertical move), per column
    int best = 0;
    for (int i = 1; i <= la; ++i) {
        int diag = 0, F = -(1 << 28);    // F: gap in b (horizontal move) along the row
        w.H[0] = 0;
        for (int j = 1; j <= lb; ++j) {
            const int up = w.H[j];
            w.E[j] = std::max(w.E[j] + ext, up + open);
            F = std::max(F + ext, w.H[j - 1] + open);
            int h = diag + sub_score(sub, (unsigned char)a[i - 1], (unsigned char)b[j - 1]);
            h = std::max(h, std::max(w.E[j], F));
            h = std::max(h, 0);
            diag = up;
            w.H[j] = h;
            best = std::max(best, h);
        }
    }
    return best;
}
}  // namespace

extern "C" int dca_sw_scores(const char* ref, int lref, const char* seqs, const int* offsets, int nseq, const int* sub,
                             int gap_open, int gap_extend, int* scores_out)
{
    if (!ref || !seqs || !offsets || !sub || !scores_out || lref < 0 || nseq < 0) return DCA_ERR_ARG;
    SwRows w;
    for (int k = 0; k < nseq; ++k)
        scores_out[k] = sw_score_one(ref, lref, seqs + offsets[k], offsets[k + 1] - offsets[k], sub, gap_open, gap_extend, w);
    return DCA_OK;
}

// aligned_a / aligned_b: capacity la + lb + 1 each; receive the aligned REGION only (with '-').
// start_a / start_b: 0-based index of the first residue of the region in each sequence.
extern "C" int dca_sw_align(const char* a, int la, const char* b, int lb, const int* sub, int gap_open, int gap_extend,
                            int* score_out, int* start_a, int* start_b, char* aligned_a, char* aligned_b, int* aligned_len)
{
    if (!a || !b || !sub || !score_out || !start_a || !start_b || !aligned_a || !aligned_b || !aligned_len) return DCA_ERR_ARG;
    const int W = lb + 1;
    const int NEG = -(1 << 28);
    std::vector<int> H((size_t)(la + 1) * W, 0), E((size_t)(la + 1) * W, NEG), F((size_t)(la + 1) * W, NEG);
    int best = 0, bi = 0, bj = 0;
    for (int i = 1; i <= la; ++i)
        for (int j = 1; j <= lb; ++j) {
            const size_t c = (size_t)i * W + j;
            E[c] = std::max(E[c - W] + gap_extend, H[c - W] + gap_open);     // gap in b's row direction: consumes a[i-1]
            F[c] = std::max(F[c - 1] + gap_extend, H[c - 1] + gap_open);     // consumes b[j-1]
            int h = H[c - W - 1] + sub_score(sub, (unsigned char)a[i - 1], (unsigned char)b[j - 1]);
            h = std::max(h, std::max(E[c], F[c]));
            h = std::max(h, 0);
            H[c] = h;
            if (h > best) { best = h; bi = i; bj = j; }
        }
    *score_out = best;
    std::string ra, rb;
    int i = bi, j = bj, state = 0;    // 0: in H, 1: in E (gap in b), 2: in F (gap in a)
    while (i > 0 && j > 0) {
        const size_t c = (size_t)i * W + j;
        if (state == 0) {
            if (H[c] == 0) break;
            if (H[c] == H[c - W - 1] + sub_score(sub, (unsigned char)a[i - 1], (unsigned char)b[j - 1])) {
                ra.push_back(a[i - 1]); rb.push_back(b[j - 1]); --i; --j;
            } else if (H[c] == E[c]) state = 1;
            else state = 2;
        } else if (state == 1) {
            ra.push_back(a[i - 1]); rb.push_back('-');
            if (E[c] == H[c - W] + gap_open) state = 0;
            --i;
        } else {
            ra.push_back('-'); rb.push_back(b[j - 1]);
            if (F[c] == H[c - 1] + gap_open) state = 0;
            --j;
        }
    }
    *start_a = i;
    *start_b = j;
    const int n = (int)ra.size();
    for (int k = 0; k < n; ++k) { aligned_a[k] = ra[n - 1 - k]; aligned_b[k] = rb[n - 1 - k]; }
    aligned_a[n] = aligned_b[n] = 0;
    *aligned_len = n;
    return DCA_OK;
}
