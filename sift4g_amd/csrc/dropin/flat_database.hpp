// flat_database.hpp -- what the device-prefilter build's two replaced stages share: the database as the prefilter pass read it.
//
// The reference reads the FASTA file twice -- searchDatabase (database_search.cpp:80-97) and alignDatabase
// (database_alignment.cpp:36-47) -- and builds every record into a Chain both times.  device_database_search.cpp reads it as flat
// residue codes (nothing else goes to the device) and, instead of freeing each part, leaves it here with its record names;
// batched_database_alignment.cpp finds the parts and aligns straight out of them (chainDatabaseCreateFlat): one pass over the
// file, and a Chain object only for the records that end up in an alignment.  Beyond S4G_FLAT_DB_BYTES (default 16 GB) of
// residues + names the first stage stops keeping parts and the second reads the file again, as before.
#pragma once
#include <cstdlib>
#include <string>
#include <vector>

namespace s4g_dropin {

struct FlatPart {
    char* codes = nullptr;          // malloc'ed by readFastaCodesPartNamed, as are the other three
    size_t* off = nullptr;
    char* names = nullptr;
    size_t* name_off = nullptr;
    int n = 0;
    unsigned first_id = 0;
};

struct FlatDatabase {
    std::string path;
    std::vector<FlatPart> parts;
    bool complete = false;          // every part of `path` is here
    size_t bytes = 0;
    void clear() {
        for (FlatPart& p : parts) { free(p.codes); free(p.off); free(p.names); free(p.name_off); }
        parts.clear(); complete = false; bytes = 0; path.clear();
    }
};

inline FlatDatabase& flat_database() { static FlatDatabase db; return db; }

inline size_t flat_database_budget() {
    const char* v = getenv("S4G_FLAT_DB_BYTES");
    const long long n = v ? atoll(v) : -1;
    return n >= 0 ? (size_t)n : (size_t)16 << 30;
}

}  // namespace s4g_dropin
