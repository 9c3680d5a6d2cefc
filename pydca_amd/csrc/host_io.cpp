// Host-side MSA reader with the semantics of PlmDCA::readSequencesFromFile
// (pydca/plmdca/plmdca_numerics.cpp:685-767): every non-empty line not starting with '>'
// is one sequence, only its first L characters are used (upper-cased), residues map to
// 0-based codes with gap = q-1, rows already seen are dropped (first occurrence kept).
// Where the reference throws (unopenable file :743-746, character missing from its table
// :752) this returns an error code instead.
#include <cctype>
#include <fstream>
#include <string>
#include <unordered_set>

#include "dca_internal.h"

namespace {

// residue tables of plmdca_numerics.cpp:708-717 (protein) and :722-731 (RNA; no 'T')
struct CodeTable {
    int8_t protein[256];
    int8_t rna[256];
    CodeTable()
    {
        for (int c = 0; c < 256; ++c) protein[c] = rna[c] = -1;
        const char* aa = "ACDEFGHIKLMNPQRSTVWY";
        for (int k = 0; aa[k]; ++k) protein[(unsigned char)aa[k]] = (int8_t)k;
        for (const char* p = "-.~BJOUXZ"; *p; ++p) protein[(unsigned char)*p] = 20;
        rna[(unsigned char)'A'] = 0; rna[(unsigned char)'C'] = 1; rna[(unsigned char)'G'] = 2; rna[(unsigned char)'U'] = 3;
        for (const char* p = "-~.BDEFHIJKLMNOPQRSVWXYZ"; *p; ++p) rna[(unsigned char)*p] = 4;
    }
};
const CodeTable kCodes;

}  // namespace

int dca_read_msa_impl(const char* path, int biomolecule, int L, uint8_t* out, int capacity, int* raw_count)
{
    if (!path || L <= 0 || (biomolecule != DCA_BIOMOLECULE_PROTEIN && biomolecule != DCA_BIOMOLECULE_RNA)) {
        dca_set_error("dca_read_msa: bad arguments");
        return DCA_ERR_ARG;
    }
    std::ifstream in(path);
    if (in.fail()) {
        dca_set_error("Unable to open file %s", path);
        return DCA_ERR_IO;
    }
    const int8_t* table = biomolecule == DCA_BIOMOLECULE_PROTEIN ? kCodes.protein : kCodes.rna;
    std::unordered_set<std::string> seen;
    std::string line, row((size_t)L, '\0');
    int nuniq = 0, nraw = 0;
    while (std::getline(in, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        if (line.empty() || line[0] == '>') continue;
        if ((int)line.size() < L) {
            dca_set_error("sequence line %d of %s is shorter than %d", nraw + 1, path, L);
            return DCA_ERR_RESIDUE;
        }
        for (int s = 0; s < L; ++s) {
            const int code = table[(unsigned char)std::toupper((unsigned char)line[s])];
            if (code < 0) {
                dca_set_error("character '%c' of sequence line %d is not in the residue table", line[s], nraw + 1);
                return DCA_ERR_RESIDUE;
            }
            row[s] = (char)code;
        }
        ++nraw;
        if (!seen.insert(row).second) continue;
        if (out) {
            if (nuniq >= capacity) {
                dca_set_error("dca_read_msa: capacity %d too small", capacity);
                return DCA_ERR_ARG;
            }
            memcpy(out + (size_t)nuniq * L, row.data(), (size_t)L);
        }
        ++nuniq;
    }
    if (raw_count) *raw_count = nraw;
    return nuniq;
}
