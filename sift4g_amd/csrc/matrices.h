// matrices.h -- substitution tables known to scorerCreateMatrix by name
// (/root/reference/sift4g/src/main.cpp:73,208; the names accepted by the CLI
// are listed in its help text, main.cpp:304-314).
//
// The tables are the published NCBI matrices (24 symbols) re-indexed to the 26
// letter codes (code = letter - 'A') the swsharp API hands to sift4g
// (/root/reference/sift4g/src/hash.cpp:16-19).  Letters without an NCBI row
// (J, O, U) score as X.  This re-indexing policy is an ASSUMPTION: swsharp's own
// table is in an absent submodule and cannot be compared.
#pragma once
#include <cstdint>
#include <cstring>

namespace s4g {

// NCBI BLOSUM62, symbol order ARNDCQEGHILKMFPSTWYVBZX*, one row per line.
static const char* const kNcbiOrder = "ARNDCQEGHILKMFPSTWYVBZX*";
static const char* const kBlosum62Text =
    " 4 -1 -2 -2  0 -1 -1  0 -2 -1 -1 -1 -1 -2 -1  1  0 -3 -2  0 -2 -1  0 -4\n"
    "-1  5  0 -2 -3  1  0 -2  0 -3 -2  2 -1 -3 -2 -1 -1 -3 -2 -3 -1  0 -1 -4\n"
    "-2  0  6  1 -3  0  0  0  1 -3 -3  0 -2 -3 -2  1  0 -4 -2 -3  3  0 -1 -4\n"
    "-2 -2  1  6 -3  0  2 -1 -1 -3 -4 -1 -3 -3 -1  0 -1 -4 -3 -3  4  1 -1 -4\n"
    " 0 -3 -3 -3  9 -3 -4 -3 -3 -1 -1 -3 -1 -2 -3 -1 -1 -2 -2 -1 -3 -3 -2 -4\n"
    "-1  1  0  0 -3  5  2 -2  0 -3 -2  1  0 -3 -1  0 -1 -2 -1 -2  0  3 -1 -4\n"
    "-1  0  0  2 -4  2  5 -2  0 -3 -3  1 -2 -3 -1  0 -1 -3 -2 -2  1  4 -1 -4\n"
    " 0 -2  0 -1 -3 -2 -2  6 -2 -4 -4 -2 -3 -3 -2  0 -2 -2 -3 -3 -1 -2 -1 -4\n"
    "-2  0  1 -1 -3  0  0 -2  8 -3 -3 -1 -2 -1 -2 -1 -2 -2  2 -3  0  0 -1 -4\n"
    "-1 -3 -3 -3 -1 -3 -3 -4 -3  4  2 -3  1  0 -3 -2 -1 -3 -1  3 -3 -3 -1 -4\n"
    "-1 -2 -3 -4 -1 -2 -3 -4 -3  2  4 -2  2  0 -3 -2 -1 -2 -1  1 -4 -3 -1 -4\n"
    "-1  2  0 -1 -3  1  1 -2 -1 -3 -2  5 -1 -3 -1  0 -1 -3 -2 -2  0  1 -1 -4\n"
    "-1 -1 -2 -3 -1  0 -2 -3 -2  1  2 -1  5  0 -2 -1 -1 -1 -1  1 -3 -1 -1 -4\n"
    "-2 -3 -3 -3 -2 -3 -3 -3 -1  0  0 -3  0  6 -4 -2 -2  1  3 -1 -3 -3 -1 -4\n"
    "-1 -2 -2 -1 -3 -1 -1 -2 -2 -3 -3 -1 -2 -4  7 -1 -1 -4 -3 -2 -2 -1 -2 -4\n"
    " 1 -1  1  0 -1  0  0  0 -1 -2 -2  0 -1 -2 -1  4  1 -3 -2 -2  0  0  0 -4\n"
    " 0 -1  0 -1 -1 -1 -1 -2 -2 -1 -1 -1 -1 -2 -1  1  5 -2 -2  0 -1 -1  0 -4\n"
    "-3 -3 -4 -4 -2 -2 -3 -2 -2 -3 -2 -3 -1  1 -4 -3 -2 11  2 -3 -4 -3 -2 -4\n"
    "-2 -2 -2 -3 -2 -1 -2 -3  2 -1 -1 -2 -1  3 -3 -2 -2  2  7 -1 -3 -2 -1 -4\n"
    " 0 -3 -3 -3 -1 -2 -2 -3 -3  3  1 -2  1 -1 -2 -2  0 -3 -1  4 -3 -2 -1 -4\n"
    "-2 -1  3  4 -3  0  1 -1  0 -3 -4  0 -3 -3 -2  0 -1 -4 -3 -3  4  1 -1 -4\n"
    "-1  0  0  1 -3  3  4 -2  0 -3 -3  1 -1 -3 -1  0 -1 -3 -2 -2  1  4 -1 -4\n"
    " 0 -1 -1 -1 -2 -1 -1 -1 -1 -1 -1 -1 -1 -1 -2  0  0 -2 -1 -1 -1 -1 -1 -4\n"
    "-4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4  1\n";

// Parses a 24x24 NCBI text table into the letter-indexed 26x26 form. Returns false on malformed text.
inline bool ncbi_to_letters(const char* text, int32_t out[26 * 26]) {
    int v[24 * 24];
    int n = 0;
    const char* p = text;
    while (*p && n < 24 * 24) {
        while (*p == ' ' || *p == '\n') ++p;
        if (!*p) break;
        int sign = 1;
        if (*p == '-') { sign = -1; ++p; }
        if (*p < '0' || *p > '9') return false;
        int x = 0;
        while (*p >= '0' && *p <= '9') x = x * 10 + (*p++ - '0');
        v[n++] = sign * x;
    }
    if (n != 24 * 24) return false;
    int map[26];
    for (int c = 0; c < 26; ++c) {
        const char* q = strchr(kNcbiOrder, 'A' + c);
        map[c] = q ? (int)(q - kNcbiOrder) : 22;  // J, O, U -> X
    }
    for (int a = 0; a < 26; ++a)
        for (int b = 0; b < 26; ++b) out[a * 26 + b] = v[map[a] * 24 + map[b]];
    return true;
}

inline bool matrix_by_name(const char* name, int32_t out[26 * 26]) {
    if (name && strcmp(name, "BLOSUM_62") == 0) return ncbi_to_letters(kBlosum62Text, out);
    return false;
}

}  // namespace s4g
