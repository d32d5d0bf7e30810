// tools/reader_ring_bench.cc -- the node-level reader of `bptrain gpu_used=N` (csrc/host/chunk_ring.h) WITHOUT a GPU: how many raw
// frames per second the shared chunk ring delivers to N forked ranks (reference: one host reader feeding G devices,
// Interface.cc:689-861).  The parent opens the Pfiles, plans, maps the ring and forks N-1 ranks exactly as bptrain does; every rank runs
// the producer side on a helper thread (rank 0: tables + lrand48 shuffle + noise-aware rows; everyone: 1/N of the frame conversion)
// and a STAND-IN consumer on its main thread that does the host work bptrain does per chunk -- its rows of every global minibatch
// out of the shared tables -- and then either nothing more (drain=tables: the upload itself is a DMA transfer out of the pinned
// ring) or a memcpy of the raw chunk into a private buffer (drain=copy: an upper bound, the host memory traffic of a staged copy).
//
//   reader_ring_bench <fea> <targ> <norm> <fea_dim> <ctx> <targ_offset> <out_dim> <traincache> <input_dim> <sent_st> <sent_en> <seed>
//                     <world> <global_bunch> <drain: tables|copy> <passes>
// Build: g++ -O3 -std=c++17 -pthread tools/reader_ring_bench.cc dnn-for-speech-enhancement_amd/csrc/host/pfile_reader.cpp -o tools/bin/reader_ring_bench
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <thread>
#include <time.h>
#include <unistd.h>
#include <vector>

#include "../dnn-for-speech-enhancement_amd/csrc/host/chunk_ring.h"
#include "../dnn-for-speech-enhancement_amd/csrc/host/pfile_reader.h"

static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
    if (argc != 17) { fprintf(stderr, "usage: see the header of tools/reader_ring_bench.cc\n"); return 2; }
    bp::ReaderConfig rc;
    rc.fea_file = argv[1]; rc.targ_file = argv[2]; rc.norm_file = argv[3];
    rc.fea_dim = atoi(argv[4]); rc.fea_context = atoi(argv[5]); rc.targ_offset = atoi(argv[6]); rc.out_dim = atoi(argv[7]);
    rc.traincache = atoi(argv[8]); rc.input_dim = atoi(argv[9]);
    const int st = atoi(argv[10]), en = atoi(argv[11]), world = atoi(argv[13]), Bg = atoi(argv[14]), passes = atoi(argv[16]);
    const bool copy = !strcmp(argv[15], "copy");
    srand48(atoi(argv[12]));
    bp::PfileReader r(rc);
    r.open();
    const bp::PfileReader::Plan p = r.plan(st, en);
    const int nch1 = (int)p.chunk_frame_st.size(), nch = nch1 * passes;
    int fcap = 1; long samples = 0, frames = 0;
    for (int c = 0; c < nch1; ++c) { const bp::PfileReader::ChunkShape s = r.chunk_shape(p, c); if (s.n_frames > fcap) fcap = s.n_frames; samples += s.n_samples; frames += s.n_frames; }
    bp::ChunkRing ring(world, fcap, rc.traincache, en - st + 2, rc.fea_dim, rc.out_dim, r.nat());
    // per-rank results through a small shared block
    double *res = (double *)mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    int rank = 0;
    std::vector<pid_t> kids;
    for (int k = 1; k < world; ++k) { const pid_t c = fork(); if (c == 0) { rank = k; kids.clear(); break; } kids.push_back(c); }
    ring.register_rank(rank);
    { const unsigned hw = std::thread::hardware_concurrency(); r.set_convert_threads(hw > (unsigned)world ? (int)(hw / (unsigned)world) : 1); }   // as bptrain does
    const double t0 = now();
    std::thread helper([&] { for (int i = 0; i < nch; ++i) if (!ring.produce(r, p, i, i % nch1, true, rank)) return; });
    std::vector<float> priv;
    if (copy) priv.resize((size_t)fcap * (rc.fea_dim + rc.out_dim));
    double wait_s = 0, work_s = 0; long my_rows = 0;
    for (int i = 0; i < nch; ++i) {
        bp::ChunkRing::View v;
        const double a = now();
        if (!ring.acquire(i, v)) { helper.join(); fprintf(stderr, "rank %d: ring aborted: %s\n", rank, ring.error().c_str()); if (rank != 0) _exit(5); return 5; }
        const double b = now();
        const std::vector<int> rows = bp::shard_rows(v.n_samples, Bg, world, rank);
        std::vector<int> ws(rows.size()), tf(rows.size()), nr(v.nat_row ? rows.size() : 0);
        for (size_t k = 0; k < rows.size(); ++k) {
            ws[k] = v.win_start[rows[k]]; tf[k] = v.targ_frame[rows[k]];
            if (v.nat_row) nr[k] = v.nat_row[rows[k]];
        }
        if (copy) {
            memcpy(priv.data(), v.fea, (size_t)v.n_frames * rc.fea_dim * 4);
            memcpy(priv.data() + (size_t)v.n_frames * rc.fea_dim, v.targ, (size_t)v.n_frames * rc.out_dim * 4);
        }
        my_rows += (long)rows.size() + (ws.empty() ? 0 : (ws[0] & 0));
        ring.done(i);
        const double c = now();
        wait_s += b - a; work_s += c - b;
    }
    helper.join();
    const double t1 = now();
    res[16 * rank + 0] = t1 - t0; res[16 * rank + 1] = wait_s; res[16 * rank + 2] = work_s; res[16 * rank + 3] = (double)my_rows;
    { const bp::ChunkRing::Times &tt = ring.times();
      res[16 * rank + 4] = tt.wait_slot; res[16 * rank + 5] = tt.tables; res[16 * rank + 6] = tt.wait_tables; res[16 * rank + 7] = tt.convert;
      res[16 * rank + 8] = tt.wait_converted; res[16 * rank + 9] = tt.nat; }
    if (rank != 0) _exit(0);
    int bad = 0;
    for (pid_t c : kids) { int stt = 0; waitpid(c, &stt, 0); if (!WIFEXITED(stt) || WEXITSTATUS(stt) != 0) bad = 1; }
    if (bad) { fprintf(stderr, "a rank failed\n"); return 6; }
    double wall = 0;
    for (int k = 0; k < world; ++k) if (res[16 * k] > wall) wall = res[16 * k];
    printf("world %d drain=%s: %d chunks (%ld samples, %ld raw frames per pass x %d passes) in %.3f s -> %.3f M samples/s per node, %.3f M raw frames/s converted\n",
           world, argv[15], nch, samples, frames, passes, wall, samples * passes / wall * 1e-6, frames * passes / wall * 1e-6);
    for (int k = 0; k < world; ++k)
        printf("    rank %d: %.3f s total | consumer: %.3f s waiting for chunks, %.3f s work, %.0f rows (%.3f M rows/s) | producer thread: slot %.3f tables %.3f wait-tables %.3f convert %.3f wait-converted %.3f nat %.3f\n",
               k, res[16 * k], res[16 * k + 1], res[16 * k + 2], res[16 * k + 3], res[16 * k + 3] / res[16 * k] * 1e-6, res[16 * k + 4], res[16 * k + 5], res[16 * k + 6],
               res[16 * k + 7], res[16 * k + 8], res[16 * k + 9]);
    return 0;
}
