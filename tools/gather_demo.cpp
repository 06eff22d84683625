// gather_demo.cpp -- the one-process-per-GPU layout of BASELINE's north_star in plain C++ over the C ABI (include/s4g_hip.h), no
// Python, no torch: N ranks, each on its own MI355X, scan their share and meet once in s4g_gather_topk (RCCL all-gather over xGMI +
// merge on the device).  Self-checking: rank 0 also scans the whole job alone and compares the gathered lists with that.
//
//   build/gather_demo --ranks N [--mode query-shard|db-shard] [--queries Q] [--targets T] [--k K]
//
// The parent process touches no GPU: it starts the N ranks as child processes (fork + exec of itself with --rank r), hands them
// a file name through which rank 0 publishes the communicator id (s4g_comm_unique_id), waits, and exits with the worst exit code.
// Rank r runs on device r (S4G_DEMO_DEVICES="0,0" maps ranks to devices by hand; RCCL itself refuses two ranks on one device).
//
// What the reference does in its place: `--cards 0123` hands the card list to alignDatabase, whose threads share one process
// (/root/reference/sift4g/src/main.cpp:254-262, database_alignment.cpp:83-86) and whose chunk results meet in dbAlignmentsMerge
// on the host (:97-104).
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "s4g_hip.h"

#define CHECK(expr)                                                                             \
    do {                                                                                        \
        int rc_ = (expr);                                                                       \
        if (rc_ != S4G_OK) { fprintf(stderr, "[rank %d] %s failed (%d): %s\n", g_rank, #expr, rc_, s4g_last_error()); exit(3); } \
    } while (0)
#define HIPCHECK(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) { fprintf(stderr, "[rank %d] %s failed: %s\n", g_rank, #expr, hipGetErrorString(e_)); exit(3); } \
    } while (0)

static int g_rank = -1;

static uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// n sequences of lengths in [lo, hi] over the 20 standard residues; every `plant`-th one starts with a copy of a query
static void make_set(uint64_t seed, uint32_t n, uint32_t lo, uint32_t hi, std::vector<uint8_t>& codes, std::vector<uint64_t>& off) {
    static const uint8_t std20[20] = {0, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 15, 16, 17, 18, 19, 21, 22, 24};
    off.assign(n + 1, 0);
    uint64_t s = seed;
    for (uint32_t i = 0; i < n; ++i) off[i + 1] = off[i] + lo + (uint32_t)(splitmix(s) % (hi - lo + 1));
    codes.resize(off[n]);
    for (uint64_t i = 0; i < off[n]; ++i) codes[i] = std20[splitmix(s) % 20];
}

struct Lists { std::vector<uint32_t> ids, cnt; std::vector<int32_t> sc; };

static Lists download(const uint32_t* d_ids, const int32_t* d_sc, const uint32_t* d_cnt, uint32_t nq, uint32_t k) {
    Lists l;
    l.ids.resize((size_t)nq * k); l.sc.resize((size_t)nq * k); l.cnt.resize(nq);
    HIPCHECK(hipMemcpy(l.ids.data(), d_ids, l.ids.size() * 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(l.sc.data(), d_sc, l.sc.size() * 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(l.cnt.data(), d_cnt, l.cnt.size() * 4, hipMemcpyDeviceToHost));
    return l;
}

static int run_rank(int rank, int world, const std::string& id_file, const std::string& mode, uint32_t nq, uint32_t nt, uint32_t k) {
    g_rank = rank;
    int device = rank;
    if (const char* map = getenv("S4G_DEMO_DEVICES")) {           // "0,1,0,1": rank -> device
        std::vector<int> d;
        for (const char* p = map; *p;) { d.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p) ++p; }
        if (rank < (int)d.size()) device = d[rank];
    }
    int n_dev = 0;
    CHECK(s4g_device_count(&n_dev));
    if (device >= n_dev) { fprintf(stderr, "[rank %d] device %d of %d: not enough GPUs for this many ranks\n", rank, device, n_dev); return 4; }
    HIPCHECK(hipSetDevice(device));

    // ---- the communicator: rank 0 makes the id and publishes it through the file the parent named ----
    uint8_t id[128];
    if (rank == 0) {
        CHECK(s4g_comm_unique_id(id));
        const std::string tmp = id_file + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(id, 1, 128, f) != 128) { fprintf(stderr, "[rank 0] cannot write %s\n", tmp.c_str()); return 3; }
        fclose(f);
        rename(tmp.c_str(), id_file.c_str());
    } else {
        FILE* f = NULL;
        for (int tries = 0; tries < 6000 && !(f = fopen(id_file.c_str(), "rb")); ++tries) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        if (!f || fread(id, 1, 128, f) != 128) { fprintf(stderr, "[rank %d] no communicator id in %s\n", rank, id_file.c_str()); return 3; }
        fclose(f);
    }
    s4g_comm* comm = NULL;
    CHECK(s4g_comm_create(&comm, device, rank, world, id));
    int rccl_version = 0;
    CHECK(s4g_comm_info(comm, NULL, NULL, NULL, &rccl_version));

    // ---- the same synthetic job on every rank (deterministic generator), a copy of query i planted in target 7 i + 3 ----
    std::vector<uint8_t> q_codes, t_codes;
    std::vector<uint64_t> q_off, t_off;
    make_set(0x51F74600ull + 11, nq, 300, 400, q_codes, q_off);
    make_set(0x51F74600ull + 12, nt, 420, 900, t_codes, t_off);
    for (uint32_t i = 0; i < nq && 7 * i + 3 < nt; ++i)
        memcpy(&t_codes[t_off[7 * i + 3]], &q_codes[q_off[i]], q_off[i + 1] - q_off[i]);
    int32_t table[26 * 26];
    CHECK(s4g_matrix_by_name("BLOSUM_62", table));
    s4g_scorer* scorer = NULL;
    CHECK(s4g_scorer_create(&scorer, table, 10, 1));
    s4g_queries* queries = NULL;
    CHECK(s4g_queries_create(&queries, device, q_codes.data(), q_off.data(), nq));

    uint32_t *d_ids = NULL, *d_cnt = NULL, *g_ids = NULL, *g_cnt = NULL;
    int32_t *d_sc = NULL, *g_sc = NULL;
    HIPCHECK(hipMalloc((void**)&d_ids, (size_t)nq * k * 4)); HIPCHECK(hipMalloc((void**)&d_sc, (size_t)nq * k * 4)); HIPCHECK(hipMalloc((void**)&d_cnt, (size_t)nq * 4));
    HIPCHECK(hipMalloc((void**)&g_ids, (size_t)nq * k * 4)); HIPCHECK(hipMalloc((void**)&g_sc, (size_t)nq * k * 4)); HIPCHECK(hipMalloc((void**)&g_cnt, (size_t)nq * 4));
    hipStream_t stream;
    HIPCHECK(hipStreamCreate(&stream));
    const auto t0 = std::chrono::steady_clock::now();
    s4g_db* db = NULL;
    uint64_t cells = 0;
    if (mode == "query-shard") {
        // database replicated; this rank scores queries [b0, b1)
        CHECK(s4g_db_create(&db, device, t_codes.data(), t_off.data(), nt));
        const uint32_t b0 = (uint32_t)((uint64_t)nq * rank / world), b1 = (uint32_t)((uint64_t)nq * (rank + 1) / world);
        if (b1 > b0) {
            CHECK(s4g_scan_topk(db, queries, b0, b1 - b0, scorer, k, 1, d_ids, d_sc, d_cnt, stream, 0));
            CHECK(s4g_scan_cells(db, queries, b0, b1 - b0, &cells));
        }
        CHECK(s4g_gather_topk(comm, S4G_GATHER_QUERY_SHARD, nq, k, d_ids, d_sc, d_cnt, 0, NULL, k, g_ids, g_sc, g_cnt, stream));
    } else {
        // contiguous slices of the database; every rank scores all queries against its own
        const uint32_t t0i = (uint32_t)((uint64_t)nt * rank / world), t1i = (uint32_t)((uint64_t)nt * (rank + 1) / world);
        std::vector<uint64_t> off(t1i - t0i + 1);
        for (uint32_t i = t0i; i <= t1i; ++i) off[i - t0i] = t_off[i] - t_off[t0i];
        CHECK(s4g_db_create(&db, device, t_codes.data() + t_off[t0i], off.data(), t1i - t0i));
        CHECK(s4g_scan_topk(db, queries, 0, nq, scorer, k, 1, d_ids, d_sc, d_cnt, stream, 0));
        CHECK(s4g_scan_cells(db, queries, 0, nq, &cells));
        CHECK(s4g_gather_topk(comm, S4G_GATHER_DB_SHARD, nq, k, d_ids, d_sc, d_cnt, t0i, NULL, k, g_ids, g_sc, g_cnt, stream));
    }
    HIPCHECK(hipStreamSynchronize(stream));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    Lists got = download(g_ids, g_sc, g_cnt, nq, k);
    fprintf(stderr, "[rank %d of %d, device %d, RCCL %d] %s: %.3e cells + gather in %.3f s (incl. database upload)\n", rank, world, device,
            rccl_version, mode.c_str(), (double)cells, dt);

    int bad = 0;
    if (rank == 0) {
        // the whole job on this one GPU, for comparison
        s4g_db* whole = NULL;
        CHECK(s4g_db_create(&whole, device, t_codes.data(), t_off.data(), nt));
        CHECK(s4g_scan_topk(whole, queries, 0, nq, scorer, k, 1, d_ids, d_sc, d_cnt, stream, 0));
        HIPCHECK(hipStreamSynchronize(stream));
        Lists want = download(d_ids, d_sc, d_cnt, nq, k);
        for (uint32_t q = 0; q < nq; ++q) {
            if (got.cnt[q] != want.cnt[q]) ++bad;
            for (uint32_t j = 0; j < k; ++j)
                if (got.ids[(size_t)q * k + j] != want.ids[(size_t)q * k + j] || got.sc[(size_t)q * k + j] != want.sc[(size_t)q * k + j]) ++bad;
            if (7 * q + 3 < nt && got.ids[(size_t)q * k] != 7 * q + 3) ++bad;           // the planted copy leads its list
        }
        CHECK(s4g_db_destroy(whole));
        printf("gather_demo %s ranks=%d queries=%u targets=%u k=%u rccl=%d: %s (%d mismatches against one GPU scanning everything)\n", mode.c_str(),
               world, nq, nt, k, rccl_version, bad ? "FAILED" : "OK", bad);
    }
    CHECK(s4g_comm_destroy(comm));
    CHECK(s4g_db_destroy(db));
    CHECK(s4g_queries_destroy(queries));
    CHECK(s4g_scorer_destroy(scorer));
    return bad ? 5 : 0;
}

int main(int argc, char** argv) {
    int ranks = 1, rank = -1;
    std::string mode = "db-shard", id_file;
    uint32_t nq = 64, nt = 20000, k = 50;
    for (int i = 1; i < argc; ++i) {
        auto val = [&](const char* name) -> const char* { if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", name); exit(2); } return argv[++i]; };
        if (!strcmp(argv[i], "--ranks")) ranks = atoi(val("--ranks"));
        else if (!strcmp(argv[i], "--rank")) rank = atoi(val("--rank"));
        else if (!strcmp(argv[i], "--id-file")) id_file = val("--id-file");
        else if (!strcmp(argv[i], "--mode")) mode = val("--mode");
        else if (!strcmp(argv[i], "--queries")) nq = (uint32_t)atoi(val("--queries"));
        else if (!strcmp(argv[i], "--targets")) nt = (uint32_t)atoi(val("--targets"));
        else if (!strcmp(argv[i], "--k")) k = (uint32_t)atoi(val("--k"));
        else { fprintf(stderr, "usage: %s --ranks N [--mode query-shard|db-shard] [--queries Q] [--targets T] [--k K]\n", argv[0]); return 2; }
    }
    if (mode != "query-shard" && mode != "db-shard") { fprintf(stderr, "--mode must be query-shard or db-shard\n"); return 2; }
    if (ranks < 1 || nq < 1 || nt < (uint32_t)ranks || k < 1) { fprintf(stderr, "bad sizes\n"); return 2; }
    if (rank >= 0) return run_rank(rank, ranks, id_file, mode, nq, nt, k);

    // ---- the parent: no GPU call in this process, ever ----
    // ... which a profiler's preloaded library breaks: under `rocprofv3 -- build/gather_demo --ranks N` the tool initialises the GPU before
    // main(), and the fork + execv below becomes an exec from a process that holds the GPU (refused on this pool: it takes the machine
    // down).  Profile ONE rank directly instead:  rocprofv3 ... -- build/gather_demo --ranks N --rank r --id-file F  (start the others by hand).
    {
        const char* pre = getenv("LD_PRELOAD");
        if (getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") || (pre && (strstr(pre, "rocprofiler") || strstr(pre, "roctracer")))) {
            fprintf(stderr, "gather_demo: a profiler library is preloaded; the launcher will not fork + exec rank processes under it.\n"
                            "Profile one rank directly: -- %s --ranks %d --rank <r> --id-file <path> --mode %s\n", argv[0], ranks, mode.c_str());
            return 2;
        }
    }
    char tmpl[] = "/tmp/s4g_gather_demo_XXXXXX";
    const int fd = mkstemp(tmpl);
    if (fd < 0) { perror("mkstemp"); return 3; }
    close(fd);
    unlink(tmpl);                                   // rank 0 creates it (rename): the others wait for it to appear
    std::vector<pid_t> kids;
    for (int r = 0; r < ranks; ++r) {
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); return 3; }
        if (pid == 0) {
            std::vector<std::string> a = {argv[0], "--ranks", std::to_string(ranks), "--rank", std::to_string(r), "--id-file", tmpl, "--mode", mode,
                                          "--queries", std::to_string(nq), "--targets", std::to_string(nt), "--k", std::to_string(k)};
            std::vector<char*> av;
            for (auto& s : a) av.push_back(const_cast<char*>(s.c_str()));
            av.push_back(NULL);
            execv(argv[0], av.data());
            perror("execv");
            _exit(127);
        }
        kids.push_back(pid);
    }
    int worst = 0;
    for (pid_t pid : kids) {
        int st = 0;
        waitpid(pid, &st, 0);
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st);
        worst = std::max(worst, code);
    }
    unlink(tmpl);
    return worst;
}
