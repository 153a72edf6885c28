// fmi — GPU driver with the CLI of R/benchmarks/fmi/fmi.cpp:  fmi <ref_file> <query_set> <batch_size> <minSeedLen> <n_threads>
// (fmi.cpp:54-58).  query_set: FASTA or FASTQ as bseq_read_one_fasta_file takes it, uncompressed (sequence lines may
// wrap in FASTA; FASTQ records are four lines).  Bases are encoded as fmi.cpp:113-124 does (A C G T -> 0 1 2 3, anything
// else 4) and every read is padded to the longest one with 4s, like the reference's enc_qdb rows.
// ref_file: as in the reference, the prefix of a bwa-mem2 index (<ref_file>.bwt.2bit.64, what `bwa-mem2 index` writes and
// FMI_search::load_index reads), or a file of genomicsbench_amd/fmi.py:save_index (read_index below).
// batch_size is accepted and ignored: the three seeding rounds only combine SMEMs of one read and batches are contiguous
// rid ranges, so the sorted output does not depend on it - all reads go to the GPU in one call.  n_threads = ingest
// threads.  --print (the reference needs a PRINT_OUTPUT rebuild): the SMEMs in the format of fmi.cpp:312-343.
// --parse-only stops after the ingest and prints counts and a checksum (no GPU needed).
#include "driver_common.h"
#include <algorithm>

// The index tables FMI_search::load_index fills (reference_seq_len, count[5], sentinel_index, cp_occ[]), from either
//   * bwa-mem2's own file <ref_file>.bwt.2bit.64 - the reference opens the index by prefix exactly like this (fmi.cpp:79-80;
//     layout as published in bwa-mem2's src/FMI_search.cpp, tools/bwa-mem2 being an empty submodule here: UNPINNED):
//     int64 reference_seq_len, int64 count[5] (load_index adds 1 to each), the CP_OCC records, the suffix-array samples
//     (int8 ms bytes then uint32 ls words; one per 8 rows from v2.1 on, one per row before: the size is taken from the file
//     length, the search never reads them), int64 sentinel_index; <ref_file> may also name such a file directly; or
//   * <ref_file> written by genomicsbench_amd/fmi.py:save_index: "GBXFMI01", int64 ref_seq_len, int64 count[5] (final
//     values), int64 sentinel_index, the CP_OCC records.
static bool read_index(const char *ref_file, gbx_fmi_index &idx, std::vector<gbx_fmi_cp_occ> &cp)
{
    const std::string pref = std::string(ref_file) + ".bwt.2bit.64";
    FILE *f = fopen(pref.c_str(), "rb");
    std::string path = pref;
    bool bwa = f != nullptr;
    if (!f) {
        f = fopen(ref_file, "rb");
        path = ref_file;
        if (!f) { fprintf(stderr, "cannot open %s or %s\n", pref.c_str(), ref_file); return false; }
        char magic[8];
        if (fread(magic, 1, 8, f) != 8) { fprintf(stderr, "%s: truncated\n", ref_file); fclose(f); return false; }
        bwa = memcmp(magic, "GBXFMI01", 8) != 0;
        if (bwa) rewind(f);
    }
    bool ok = fread(&idx.ref_seq_len, 8, 1, f) == 1 && fread(idx.count, 8, 5, f) == 5;
    if (ok && !bwa) ok = fread(&idx.sentinel_index, 8, 1, f) == 1;
    if (!ok || idx.ref_seq_len < 2 || idx.ref_seq_len > ((int64_t)1 << 40)) { fprintf(stderr, "%s: not an fmi index\n", path.c_str()); fclose(f); return false; }
    cp.resize((size_t)(idx.ref_seq_len >> 6) + 1);
    if (fread(cp.data(), sizeof(gbx_fmi_cp_occ), cp.size(), f) != cp.size()) { fprintf(stderr, "%s: truncated\n", path.c_str()); fclose(f); return false; }
    if (bwa) {
        const long at = ftell(f);
        fseek(f, 0, SEEK_END);
        const int64_t rest = (int64_t)ftell(f) - at - 8, n = idx.ref_seq_len;
        if (rest + 8 == 5 * n) {
            // one suffix-array sample per row and NO trailing sentinel_index (builds without SA compression that derive it on load):
            // it is the row whose suffix starts at 0.  The file holds the samples' upper bytes (n), then their lower words (4 n).
            std::vector<int8_t> ms((size_t)n);
            std::vector<uint32_t> ls((size_t)1 << 20);
            fseek(f, at, SEEK_SET);
            bool found = fread(ms.data(), 1, (size_t)n, f) == (size_t)n, hit = false;
            for (int64_t i = 0; found && i < n && !hit; i += (int64_t)ls.size()) {
                const size_t m = (size_t)std::min<int64_t>((int64_t)ls.size(), n - i);
                if (fread(ls.data(), 4, m, f) != m) { found = false; break; }
                for (size_t k = 0; k < m; ++k) if (ls[k] == 0 && ms[(size_t)i + k] == 0) { idx.sentinel_index = i + (int64_t)k; hit = true; break; }
            }
            if (!found || !hit) { fprintf(stderr, "%s: no suffix-array sample is 0: cannot derive sentinel_index\n", path.c_str()); fclose(f); return false; }
            for (int c = 0; c < 5; ++c) idx.count[c] += 1;
            fprintf(stderr, "index: bwa-mem2 file %s (uncompressed suffix-array samples, sentinel_index %lld derived from them)\n", path.c_str(), (long long)idx.sentinel_index);
            fclose(f);
            return true;
        }
        if (rest < 0 || rest % 5 || (rest / 5 != n && rest / 5 != (n >> 3) + 1)) {
            fprintf(stderr, "%s: %lld bytes of suffix-array samples fit neither published layout of a .bwt.2bit.64 file\n", path.c_str(), (long long)rest);
            fclose(f);
            return false;
        }
        fseek(f, -8, SEEK_END);
        if (fread(&idx.sentinel_index, 8, 1, f) != 1) { fclose(f); return false; }
        for (int c = 0; c < 5; ++c) idx.count[c] += 1;               // FMI_search::load_index
        fprintf(stderr, "index: bwa-mem2 file %s (%s suffix-array samples skipped)\n", path.c_str(), rest / 5 == n ? "uncompressed" : "1-in-8");
    }
    fclose(f);
    return true;
}

static void help() { fprintf(stderr, "Need five arguments : ref_file query_set batch_size minSeedLen n_threads [--print] [--parse-only]\n"); }

int main(int argc, char **argv)
{
    const int gpus = take_gpus_flag(argc, argv);
    std::vector<const char *> pos;
    bool print = false, parse_only = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--index-info") && i + 1 < argc) {      // fmi --index-info <ref_file>: the tables' scalars and a checksum, no GPU
            gbx_fmi_index ix;
            std::vector<gbx_fmi_cp_occ> c;
            if (!read_index(argv[i + 1], ix, c)) return EXIT_FAILURE;
            printf("{\"ref_seq_len\": %lld, \"count\": [%lld, %lld, %lld, %lld, %lld], \"sentinel_index\": %lld, \"checkpoints\": %zu, \"fnv1a\": \"%016llx\"}\n",
                   (long long)ix.ref_seq_len, (long long)ix.count[0], (long long)ix.count[1], (long long)ix.count[2], (long long)ix.count[3],
                   (long long)ix.count[4], (long long)ix.sentinel_index, c.size(), (unsigned long long)fnv1a(c.data(), c.size() * sizeof(gbx_fmi_cp_occ)));
            return 0;
        }
        if (!strcmp(argv[i], "--print")) print = true;
        else if (!strcmp(argv[i], "--parse-only")) parse_only = true;
        else pos.push_back(argv[i]);
    }
    if (pos.size() != 5) { help(); return 1; }
    const int min_seed_len = atoi(pos[3]);
    int threads = atoi(pos[4]);
    if (threads < 1) threads = 1;
    if (atoi(pos[2]) <= 0 || min_seed_len <= 0) { help(); return 1; }

    // ---- reads: records start at a line whose first byte is '>' (FASTA) or at every fourth line (FASTQ, first byte '@')
    std::vector<char> text;
    if (!slurp(pos[1], text)) { fprintf(stderr, "[E::%s] fail to open file `%s'.\n", __func__, pos[1]); return EXIT_FAILURE; }
    const double t0 = now_s();
    std::vector<const char *> line;
    std::vector<int> llen;
    if (text.size() > 1 && text[text.size() - 2] != '\n') { text[text.size() - 1] = '\n'; text.push_back(0); }   // last line without a newline
    split_lines(text.data(), text.size() - 1, threads, line, llen);
    const bool fastq = !line.empty() && line[0][0] == '@';
    std::vector<size_t> rec;                                     // first sequence line of every read; FASTA: up to the next '>'
    if (fastq) { for (size_t k = 0; k + 1 < line.size(); k += 4) rec.push_back(k + 1); }
    else for (size_t k = 0; k < line.size(); ++k) if (llen[k] > 0 && line[k][0] == '>') rec.push_back(k + 1);
    const int64_t n_reads = (int64_t)rec.size();
    std::vector<int32_t> len((size_t)n_reads, 0);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t r = 0; r < n_reads; ++r) {
        int L = 0;
        if (fastq) L = llen[rec[(size_t)r]];
        else for (size_t k = rec[(size_t)r]; k < line.size() && !(llen[k] > 0 && line[k][0] == '>'); ++k) L += llen[k];
        len[(size_t)r] = L;
    }
    int max_len = 0, min_len = n_reads ? len[0] : 0;
    for (int64_t r = 0; r < n_reads; ++r) { max_len = len[(size_t)r] > max_len ? len[(size_t)r] : max_len; min_len = len[(size_t)r] < min_len ? len[(size_t)r] : min_len; }
    if (n_reads == 0 || max_len == 0) { printf("ERROR! seqs = NULL\n"); return EXIT_FAILURE; }
    printf("numReads = %lld, max_readlength = %d, min_readlength = %d\n", (long long)n_reads, max_len, min_len);
    std::vector<uint8_t> enc((size_t)n_reads * (size_t)max_len, 4);
    std::vector<int64_t> off((size_t)n_reads);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t r = 0; r < n_reads; ++r) {
        uint8_t *q = enc.data() + (size_t)r * (size_t)max_len;
        off[(size_t)r] = r * (int64_t)max_len;                   // query_cum_len_ar (fmi.cpp:110)
        int o = 0;
        for (size_t k = rec[(size_t)r]; o < len[(size_t)r]; ++k)
            for (int c = 0; c < llen[k]; ++c) {
                const char ch = line[k][c];
                q[o++] = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
            }
    }
    fprintf(stderr, "ingest: %lld reads, %.3f s with %d threads\n", (long long)n_reads, now_s() - t0, threads);
    if (parse_only) {
        uint64_t h = fnv1a(len.data(), len.size() * sizeof(int32_t));
        h = fnv1a(enc.data(), enc.size(), h);
        printf("{\"reads\": %lld, \"max_readlength\": %d, \"bases\": %lld, \"fnv1a\": \"%016llx\"}\n", (long long)n_reads, max_len,
               (long long)enc.size(), (unsigned long long)h);
        return 0;
    }

    // ---- index tables (read_index below)
    gbx_fmi_index idx;
    std::vector<gbx_fmi_cp_occ> cp;
    if (!read_index(pos[0], idx, cp)) return EXIT_FAILURE;
    idx.cp_occ = cp.data();
    printf("reference seq len = %lld\n", (long long)idx.ref_seq_len);
    for (int c = 0; c < 5; ++c) printf("count[%d] = %lld\n", c, (long long)idx.count[c]);

    print_device_banner(gpus);
    gbx_fmi_params prm;
    gbx_fmi_default_params(&prm, min_seed_len);
    std::vector<gbx_fmi_smem> smem((size_t)n_reads * 20);        // the reference's quota per thread (fmi.cpp:183)
    std::vector<int64_t> smem_off((size_t)n_reads + 1);
    int64_t total = 0;
    const double t1 = now_s();
    int rc = gbx_fmi_smem_host(&idx, &prm, n_reads, enc.data(), (int64_t)enc.size(), off.data(), len.data(), smem.data(),
                               (int64_t)smem.size(), smem_off.data(), &total);
    if (rc == GBX_ERR_ARG && total > (int64_t)smem.size()) {     // like the reference's "realloc" (fmi.cpp:207-216)
        smem.resize((size_t)total);
        rc = gbx_fmi_smem_host(&idx, &prm, n_reads, enc.data(), (int64_t)enc.size(), off.data(), len.data(), smem.data(),
                               (int64_t)smem.size(), smem_off.data(), &total);
    }
    die_on(rc, "gbx_fmi_smem_host");
    const double dt = now_s() - t1;
    printf("Consumed: %0.4lf sec\n", dt);                        // the reference prints cycles too (rdtsc)
    printf("totalSmems = %lld\n", (long long)total);
    if (print) {                                                 // fmi.cpp:312-343
        int64_t prev_rid = -1;
        for (int64_t i = 0; i < total; ++i) {
            const gbx_fmi_smem &s = smem[(size_t)i];
            if ((int64_t)s.rid != prev_rid)
                for (int64_t j = prev_rid + 1; j <= (int64_t)s.rid; ++j) printf("%u:\n", (unsigned)j);
            prev_rid = s.rid;
            printf("[%u,%u]\n", s.m, s.n + 1);
        }
    }
    return 0;
}
