// tests/cpp/reader_dump.cc -- test utility around bp::PfileReader / bp::wts_io (host code of the "next"
// rows N1/N2).  usage:
//   reader_dump chunks <fea> <targ> <norm> <fea_dim> <ctx> <targ_offset> <out_dim> <traincache> <input_dim>
//               <sent_st> <sent_en> <shuffle 0|1> <seed> <out.bin>
//     out.bin: int32 nchunks, total_samples, chunk_frame_st[nchunks]; per chunk: int32 n, float in[n*input_dim], targ[n*out_dim]
//   reader_dump wts <in.wts> <out.wts> <numlayers> <s0> <s1> ...        (read + re-write a weights file)
//   reader_dump epoch <out.bin> name=value ...    one epoch's data path with THIS repo's host code, in the layout that
//               oracle/ref_driver.cc writes for the reference's Interface (see there): the two dumps must be identical
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>
#include "../../dnn-for-speech-enhancement_amd/csrc/host/pfile_reader.h"
#include "../../dnn-for-speech-enhancement_amd/csrc/host/wts_io.h"
#include "../../dnn-for-speech-enhancement_amd/csrc/host/chunk_ring.h"
#include <sys/wait.h>
#include <thread>

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "chunks") && argc == 16) {
        bp::ReaderConfig rc;
        rc.fea_file = argv[2]; rc.targ_file = argv[3]; rc.norm_file = argv[4];
        rc.fea_dim = atoi(argv[5]); rc.fea_context = atoi(argv[6]); rc.targ_offset = atoi(argv[7]); rc.out_dim = atoi(argv[8]);
        rc.traincache = atoi(argv[9]); rc.input_dim = atoi(argv[10]);
        const int st = atoi(argv[11]), en = atoi(argv[12]), shuffle = atoi(argv[13]);
        srand48(atoi(argv[14]));
        bp::PfileReader r(rc);
        r.open();
        const bp::PfileReader::Plan p = r.plan(st, en);
        FILE *o = fopen(argv[15], "wb");
        const int nch = (int)p.chunk_frame_st.size(), ts = (int)p.total_samples;
        fwrite(&nch, 4, 1, o); fwrite(&ts, 4, 1, o); fwrite(p.chunk_frame_st.data(), 4, nch, o);
        std::vector<float> in((size_t)rc.traincache * rc.input_dim), tg((size_t)rc.traincache * rc.out_dim);
        for (int c = 0; c < nch; ++c) {
            const int n = r.read_chunk(p, c, shuffle != 0, in.data(), tg.data());
            fwrite(&n, 4, 1, o);
            fwrite(in.data(), 4, (size_t)n * rc.input_dim, o);
            fwrite(tg.data(), 4, (size_t)n * rc.out_dim, o);
        }
        fclose(o);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "ring") && argc == 17) {
        // reader_dump ring <fea> <targ> <norm> <fea_dim> <ctx> <targ_offset> <out_dim> <traincache> <input_dim> <sent_st> <sent_en>
        //                  <seed> <world> <global_bunch> <out_prefix>
        //   the node-level shared reader of `bptrain gpu_used=N` (chunk_ring.h) WITHOUT a GPU: the parent opens, plans and
        //   maps the ring, forks world-1 ranks; every rank helps produce each chunk and writes the stacked rows of ITS
        //   shard to <out_prefix>.rank<r>: per chunk int32 n_rows, float in[n_rows*input_dim], targ[n_rows*out_dim]
        bp::ReaderConfig rc;
        rc.fea_file = argv[2]; rc.targ_file = argv[3]; rc.norm_file = argv[4];
        rc.fea_dim = atoi(argv[5]); rc.fea_context = atoi(argv[6]); rc.targ_offset = atoi(argv[7]); rc.out_dim = atoi(argv[8]);
        rc.traincache = atoi(argv[9]); rc.input_dim = atoi(argv[10]);
        const int st = atoi(argv[11]), en = atoi(argv[12]), world = atoi(argv[14]), Bg = atoi(argv[15]);
        srand48(atoi(argv[13]));
        bp::PfileReader r(rc);
        r.open();
        const bp::PfileReader::Plan p = r.plan(st, en);
        const int nch = (int)p.chunk_frame_st.size();
        int fcap = 1;
        for (int c = 0; c < nch; ++c) { const int f = r.chunk_shape(p, c).n_frames; if (f > fcap) fcap = f; }
        bp::ChunkRing ring(world, fcap, rc.traincache, en - st + 2, rc.fea_dim, rc.out_dim, r.nat());
        int rank = 0;
        std::vector<pid_t> kids;
        for (int k = 1; k < world; ++k) { const pid_t c = fork(); if (c == 0) { rank = k; kids.clear(); break; } kids.push_back(c); }
        ring.register_rank(rank);
        // RING_DIE_RANK=k: rank k dies right here and stays a zombie (its parent, rank 0, only reaps at the end) -- the others
        // must notice within seconds instead of waiting out the ring's time budget (tests/test_pfile_reader.py)
        if (getenv("RING_DIE_RANK") && rank != 0 && atoi(getenv("RING_DIE_RANK")) == rank) _exit(3);
        std::thread helper([&] { for (int i = 0; i < nch; ++i) if (!ring.produce(r, p, i, i, true, rank)) return; });
        FILE *o = fopen((std::string(argv[16]) + ".rank" + std::to_string(rank)).c_str(), "wb");
        const int D = rc.fea_dim, ctx = rc.fea_context, OD = rc.out_dim, s0 = rc.input_dim;
        for (int i = 0; i < nch; ++i) {
            bp::ChunkRing::View v;
            if (!ring.acquire(i, v)) { helper.join(); if (rank != 0) _exit(5); return 5; }
            const std::vector<int> rows = bp::shard_rows(v.n_samples, Bg, world, rank);
            const int n = (int)rows.size();
            fwrite(&n, 4, 1, o);
            std::vector<float> in((size_t)n * s0), tg((size_t)n * OD);
            for (int k = 0; k < n; ++k) {
                memcpy(&in[(size_t)k * s0], v.fea + (size_t)v.win_start[rows[k]] * D, sizeof(float) * (size_t)ctx * D);
                if (v.nat_row) memcpy(&in[(size_t)k * s0 + (size_t)ctx * D], v.nat + (size_t)v.nat_row[rows[k]] * D, sizeof(float) * D);
                memcpy(&tg[(size_t)k * OD], v.targ + (size_t)v.targ_frame[rows[k]] * OD, sizeof(float) * OD);
            }
            fwrite(in.data(), 4, in.size(), o); fwrite(tg.data(), 4, tg.size(), o);
            ring.done(i);
        }
        fclose(o);
        helper.join();
        if (rank != 0) _exit(0);
        int bad = 0;
        for (pid_t c : kids) { int stt = 0; waitpid(c, &stt, 0); if (!WIFEXITED(stt) || WEXITSTATUS(stt) != 0) bad = 1; }
        return bad ? 6 : 0;
    }
    if (argc > 1 && !strcmp(argv[1], "infer") && argc == 14) {
        // reader_dump infer <fea> <norm> <fea_dim> <ctx> <targ_offset> <traincache> <input_dim> <sent_st> <sent_en> <out.bin> x x
        //   the enhancement chunking (plan_inference): int32 nchunks, total; per chunk: int32 n, frame_st, float in[n*input_dim]
        bp::ReaderConfig rc;
        rc.fea_file = argv[2]; rc.targ_file = argv[2]; rc.norm_file = argv[3];
        rc.fea_dim = atoi(argv[4]); rc.fea_context = atoi(argv[5]); rc.targ_offset = atoi(argv[6]); rc.out_dim = rc.fea_dim;
        rc.traincache = atoi(argv[7]); rc.input_dim = atoi(argv[8]);
        bp::PfileReader r(rc);
        r.open();
        const bp::PfileReader::Plan p = r.plan_inference(atoi(argv[9]), atoi(argv[10]));
        FILE *o = fopen(argv[11], "wb");
        const int nch = (int)p.chunk_frame_st.size(), ts = (int)p.total_samples;
        fwrite(&nch, 4, 1, o); fwrite(&ts, 4, 1, o);
        std::vector<float> in((size_t)rc.traincache * rc.input_dim), tg((size_t)rc.traincache * rc.out_dim);
        for (int c = 0; c < nch; ++c) {
            const int n = r.read_chunk(p, c, false, in.data(), tg.data());
            if (n > rc.traincache) return 4;
            fwrite(&n, 4, 1, o); fwrite(&p.chunk_frame_st[c], 4, 1, o);
            fwrite(in.data(), 4, (size_t)n * rc.input_dim, o);
        }
        fclose(o);
        return 0;
    }
    if (argc > 3 && !strcmp(argv[1], "epoch")) {
        std::map<std::string, std::string> a;
        for (int i = 3; i < argc; ++i) { const char *eq = strchr(argv[i], '='); if (eq) a[std::string(argv[i], eq - argv[i])] = eq + 1; }
        auto geti = [&](const char *k) { return atoi(a[k].c_str()); };
        auto getf = [&](const char *k, float d) { return a.count(k) ? (float)atof(a[k].c_str()) : d; };
        int ls[16] = {0}, L = 0;
        { const std::string v = a["layersizes"]; size_t pos = 0;
          while (L < 10) { const size_t c = v.find(',', pos); ls[L++] = atoi(v.substr(pos, c == std::string::npos ? c : c - pos).c_str());
                           if (c == std::string::npos) break; pos = c + 1; } }
        bp::ReaderConfig rc;
        rc.fea_file = a["fea_file"]; rc.targ_file = a["targ_file"]; rc.norm_file = a["norm_file"];
        rc.fea_dim = geti("fea_dim"); rc.fea_context = geti("fea_context"); rc.targ_offset = geti("targ_offset");
        rc.out_dim = ls[L - 1]; rc.traincache = geti("traincache"); rc.input_dim = ls[0];
        FILE *o = fopen(argv[2], "wb");
        auto put_i = [&](int v) { fwrite(&v, 4, 1, o); };
        put_i(L); for (int i = 0; i < L; ++i) put_i(ls[i]);
        // Interface::Initial order: norm file, srand48, initial weights (file or random)
        std::vector<std::vector<float>> W(L), B(L);
        float *w[16] = {0}, *b[16] = {0};
        for (int i = 1; i < L; ++i) { W[i].assign((size_t)ls[i] * ls[i - 1], 0.f); B[i].assign(ls[i], 0.f); w[i] = W[i].data(); b[i] = B[i].data(); }
        bp::PfileReader r(rc);
        srand48(geti("init_randem_seed"));
        if (a["initwts_file"].empty()) bp::random_weights(L, ls, w, b, getf("init_randem_weight_min", -0.1f), getf("init_randem_weight_max", 0.1f),
                                                          getf("init_randem_bias_min", -0.1f), getf("init_randem_bias_max", 0.1f));
        else { FILE *fi = fopen(a["initwts_file"].c_str(), "rb"); const std::string err = bp::read_weights(fi, L, ls, w, b); fclose(fi);
               if (!err.empty()) { printf("%s\n", err.c_str()); return 3; } }
        r.open();
        put_i((int)r.total_frames()); put_i((int)r.total_sents());
        fwrite(r.frames_before_sent().data(), 4, r.total_sents(), o);
        auto range = [&](const char *k, int *st, int *en) { const std::string v = a[k]; const size_t d = v.find('-');
                                                            *st = atoi(v.substr(0, d).c_str()); *en = atoi(v.substr(d + 1).c_str()); };
        int st, en;
        range("train_sent_range", &st, &en);
        const bp::PfileReader::Plan tp = r.plan(st, en);
        const int n = (int)tp.chunk_frame_st.size();
        put_i(n); put_i((int)tp.total_samples); fwrite(tp.chunk_frame_st.data(), 4, n, o);
        std::vector<int> order(n);
        for (int i = 0; i < n; ++i) order[i] = i;
        bp::PfileReader::rand_index(order.data(), n);
        fwrite(order.data(), 4, n, o);
        std::vector<float> in((size_t)rc.traincache * rc.input_dim), tg((size_t)rc.traincache * rc.out_dim);
        for (int i = 0; i < n; ++i) {
            const int cnt = r.read_chunk(tp, order[i], true, in.data(), tg.data());
            put_i(cnt); fwrite(in.data(), 4, (size_t)cnt * rc.input_dim, o); fwrite(tg.data(), 4, (size_t)cnt * rc.out_dim, o);
        }
        { FILE *fw = fopen(a["outwts_file"].c_str(), "wb"); bp::write_weights(fw, L, ls, w, b); fclose(fw); }
        range("cv_sent_range", &st, &en);
        const bp::PfileReader::Plan cp = r.plan(st, en);
        const int nc = (int)cp.chunk_frame_st.size();
        put_i(nc); put_i((int)cp.total_samples); fwrite(cp.chunk_frame_st.data(), 4, nc, o);
        for (int i = 0; i < nc; ++i) {
            const int cnt = r.read_chunk(cp, i, false, in.data(), tg.data());
            put_i(cnt); fwrite(in.data(), 4, (size_t)cnt * rc.input_dim, o); fwrite(tg.data(), 4, (size_t)cnt * rc.out_dim, o);
        }
        fclose(o);
        return 0;
    }
    if (argc > 4 && !strcmp(argv[1], "wts")) {
        const int L = atoi(argv[4]);
        int ls[16] = {0};
        for (int i = 0; i < L; ++i) ls[i] = atoi(argv[5 + i]);
        std::vector<std::vector<float>> W(L), B(L);
        float *w[16] = {0}, *b[16] = {0};
        for (int i = 1; i < L; ++i) { W[i].resize((size_t)ls[i] * ls[i - 1]); B[i].resize(ls[i]); w[i] = W[i].data(); b[i] = B[i].data(); }
        FILE *fi = fopen(argv[2], "rb");
        const std::string err = bp::read_weights(fi, L, ls, w, b);
        fclose(fi);
        if (!err.empty()) { printf("%s\n", err.c_str()); return 3; }
        FILE *fo = fopen(argv[3], "wb");
        bp::write_weights(fo, L, ls, w, b);
        fclose(fo);
        return 0;
    }
    return 2;
}
