// poa — GPU driver with the CLI of R/benchmarks/poa/msa_spoa_omp.cpp:
//   poa -s <input.fasta> -t <threads> [-m N] [-x N] [-o a[,b]] [-e a[,b]] [-n N]
// Input (msa_spoa_omp.cpp:82-116): 2 lines per record; a header whose 2nd character is '0' opens a new window.
// --print writes ">Consensus_sequence\n<seq>" per window like the reference's PRINT_OUTPUT build (:281-286).
// -t = threads of the parallel ingest; --parse-only stops after it and prints counts and a checksum (no GPU needed).
// --warmup runs the whole job once untimed first (device allocations of ~10 GB can take seconds right after another
// process released its memory; with the warm-up the timed call allocates nothing).
#include <fstream>
#include "driver_common.h"

int main(int argc, char **argv)
{
    const int gpus = take_gpus_flag(argc, argv);
    std::string seq_file = "seq.fa";
    int m = 2, x = -4, o1 = -4, e1 = -2, o2 = -24, e2 = -1, threads = 1;
    bool print = false, parse_only = false, warmup = false;
    if (argc == 1) { fprintf(stderr, "usage: ./poa -s input.fasta -t <num_threads> > cons.fasta\n"); return EXIT_FAILURE; }
    for (int i = 1; i < argc; ++i) {
        char *s;
        if (!strcmp(argv[i], "-m") && i + 1 < argc) m = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-x") && i + 1 < argc) x = 0 - atoi(argv[++i]);
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) { o1 = 0 - (int)strtol(argv[++i], &s, 10); if (*s == ',') o2 = 0 - (int)strtol(s + 1, &s, 10); }
        else if (!strcmp(argv[i], "-e") && i + 1 < argc) { e1 = 0 - (int)strtol(argv[++i], &s, 10); if (*s == ',') e2 = 0 - (int)strtol(s + 1, &s, 10); }
        else if (!strcmp(argv[i], "-n") && i + 1 < argc) ++i;
        else if (!strcmp(argv[i], "-s") && i + 1 < argc) seq_file = argv[++i];
        else if (!strcmp(argv[i], "-t") && i + 1 < argc) threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--print")) print = true;
        else if (!strcmp(argv[i], "--warmup")) warmup = true;
        else if (!strcmp(argv[i], "--parse-only")) parse_only = true;
        else if (!strcmp(argv[i], "-h")) { fprintf(stderr, "usage: ./poa -s input.fasta -t <num_threads>\n"); return 0; }
    }
    if (threads < 1) threads = 1;
    gbx_poa_params P;
    gbx_poa_default_params(&P);
    P.m = (int8_t)m; P.n = (int8_t)x; P.g = (int8_t)(o1 + e1); P.e = (int8_t)e1; P.q = (int8_t)(o2 + e2); P.c = (int8_t)e2;
    std::vector<char> text;
    if (!slurp(seq_file.c_str(), text)) { fprintf(stderr, "cannot open %s\n", seq_file.c_str()); return EXIT_FAILURE; }
    // ---- parallel ingest: lines are split by all threads, the reader's state machine (readFile(): header, sequence,
    // header, ...; seq[1]=='0' on a header starts the next window) walks the line index, the sequences are copied
    // into the arena independently
    const double t_read0 = now_s();
    if (text.size() >= 2 && text[text.size() - 2] != '\n') { text.back() = '\n'; text.push_back(0); }   // unterminated last line
    std::vector<const char *> lptr; std::vector<int> llen;
    split_lines(text.data(), text.size() - 1, threads, lptr, llen);
    const size_t nl = lptr.size();
    size_t li = 0;
    bool eof = false;
    auto getl = [&](size_t &idx) -> bool { if (li < nl) { idx = li++; return true; } eof = true; return false; };
    auto opens = [&](size_t idx) { return llen[idx] > 1 && lptr[idx][1] == '0'; };
    std::vector<int64_t> win_first(1, 0);
    std::vector<size_t> seq_line;
    size_t line = 0;
    bool have = getl(line);
    while (have && !eof) {
        if (opens(line)) {
            for (;;) {
                size_t sl;
                if (!getl(sl)) { have = false; break; }
                seq_line.push_back(sl);
                if (!getl(line)) { have = false; break; }
                if (opens(line)) break;
            }
            win_first.push_back((int64_t)seq_line.size());
        } else have = getl(line);
    }
    const size_t nseq = seq_line.size();
    std::vector<int64_t> seq_off(nseq);
    std::vector<int32_t> seq_len(nseq);
    int64_t abytes = 0;
    for (size_t k = 0; k < nseq; ++k) { seq_off[k] = abytes; seq_len[k] = llen[seq_line[k]]; abytes += seq_len[k]; }
    std::string arena((size_t)abytes, '\0');
#pragma omp parallel for num_threads(threads) schedule(static)
    for (size_t k = 0; k < nseq; ++k) memcpy(&arena[(size_t)seq_off[k]], lptr[seq_line[k]], (size_t)seq_len[k]);
    const double t_read = now_s() - t_read0;
    if (parse_only) {
        uint64_t h = fnv1a(win_first.data(), win_first.size() * 8);
        h = fnv1a(seq_len.data(), nseq * 4, h); h = fnv1a(arena.data(), (size_t)abytes, h);
        printf("{\"benchmark\":\"poa\",\"windows\":%zu,\"sequences\":%zu,\"ingest_threads\":%d,\"ingest_seconds\":%.4f,\"ingest_mb_per_s\":%.1f,\"checksum\":\"%016llx\"}\n",
               win_first.size() - 1, nseq, threads, t_read, text.size() / 1e6 / t_read, (unsigned long long)h);
        return 0;
    }
    const int64_t nw = (int64_t)win_first.size() - 1, ns = (int64_t)seq_len.size();
    fprintf(stderr, "Number of batches: %lld\n", (long long)nw);
    int lmax = 1;
    for (int v : seq_len) lmax = v > lmax ? v : lmax;
    const int64_t stride = 2 * (int64_t)lmax + 64;
    std::vector<char> cons((size_t)(nw > 0 ? nw : 1) * stride);
    std::vector<int32_t> clen((size_t)nw + 1);
    arena.resize(arena.size() + 8);
    print_device_banner(gpus);
    {
        // the workspace (one slot per window in flight, ~10 GB for the 'large' input) is allocated before the timed
        // region, as the reference creates its alignment engines and graphs before it (msa_spoa_omp.cpp:184-190)
        gbx_poa_plan plan;
        die_on(gbx_poa_plan_host(nw, win_first.data(), seq_len.data(), &plan), "gbx_poa_plan_host");
        // (several devices: every shard has a plan of its own, which the untimed run below makes and caches per device)
        if (gbx_host_devices() <= 1) die_on(gbx_host_reserve(gbx_poa_workspace_bytes(&plan)), "gbx_host_reserve");
        else warmup = true;
    }
    if (warmup)      // an untimed run of the whole job first: every allocation of the timed call is then a cache hit
        die_on(gbx_poa_consensus_host(&P, nw, win_first.data(), ns, seq_off.data(), seq_len.data(), arena.data(),
                                      (int64_t)arena.size(), cons.data(), clen.data(), stride), "gbx_poa_consensus_host (warm-up)");
    const double t0 = now_s();
    die_on(gbx_poa_consensus_host(&P, nw, win_first.data(), ns, seq_off.data(), seq_len.data(), arena.data(),
                                  (int64_t)arena.size(), cons.data(), clen.data(), stride), "gbx_poa_consensus_host");
    const double dt = now_s() - t0;
    if (print)
        for (int64_t w = 0; w < nw; ++w) printf(">Consensus_sequence\n%.*s\n", clen[w], cons.data() + w * stride);
    fprintf(stderr, "Runtime: %.2f\n", dt);
    fprintf(stderr, "{\"benchmark\":\"poa\",\"windows\":%lld,\"sequences\":%lld,\"seconds\":%.6f}\n", (long long)nw, (long long)ns, dt);
    return 0;
}
