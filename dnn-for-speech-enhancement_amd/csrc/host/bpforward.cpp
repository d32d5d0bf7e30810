// bpforward.cpp -- batch enhancement ("next" row N4 of SURVEY.md 8f): noisy log-power-spectrum Pfile in, enhanced
// log-power-spectrum Pfile out.  The reference keeps its decoder as an external download (README.md:39-44); the
// forward pass itself is cv_bunch_single (BP_GPU.cu:676-773: weights scaled by keep when the net was trained with
// dropout, linear output layer), which is what bp_forward_windows runs on the MI355X.  Input side = the reference's
// reader (Interface.cc:468-1034, via csrc/host/pfile_reader.cpp): mean/variance normalisation, fea_context stacked
// frames, optional noise-aware block; the output frame of a window is the one at offset targ_offset inside it.
//
//   bpforward fea_file=noisy.pfile norm_file=x.norm initwts_file=mlp.N.wts out_file=enh.pfile layersizes=1548,2048,...,129
//             fea_dim=129 fea_context=11 targ_offset=5 sent_range=0-99 [dropoutflag=1 visible_omit=0.1 hid_omit=0.2]
//             [bunchsize=1024] [traincache=102400] [activation=relu|sigmoid] [device=0] [compute=fp32|bf16]
//
// out_file: an ICSI Pfile with the input's sentence structure; sentence s holds one record per window of that
// sentence (frame id = window start + targ_offset), layersizes[last] features each: EVERY window of every sentence,
// exactly once, in file order (chunks are cut on sentence boundaries, PfileReader::plan_inference -- the training
// planner's cuts drop ctx-1 windows each, Interface.cc:607-614, which an enhancement tool must not).  Sentences shorter
// than the context contribute no records (as in the reader).  Errors: message + exit(0), success: return 1 (reference convention).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../../include/BP_GPU.h"
#include "pfile_reader.h"
#include "wts_io.h"

static inline uint32_t be32(uint32_t v) { return __builtin_bswap32(v); }

int main(int argc, char **argv)
{
    std::string fea_file, norm_file, wts_file, out_file, range = "";
    int fea_dim = 0, ctx = 1, toff = 0, dropoutflag = 0, bunch = 1024, cache = 102400, L = 0, ls[MAXLAYER] = {0};
    int activation = 0, device = 0, compute = 0;
    float vis = 0.f, hid = 0.f;
    for (int i = 1; i < argc; ++i) {
        char *eq = strchr(argv[i], '=');
        if (!eq) { printf("Arg: %s  Format Error\n", argv[i]); exit(0); }
        const std::string k(argv[i], eq - argv[i]), v(eq + 1);
        if (k == "fea_file") fea_file = v; else if (k == "norm_file") norm_file = v; else if (k == "initwts_file") wts_file = v;
        else if (k == "out_file") out_file = v; else if (k == "sent_range") range = v;
        else if (k == "fea_dim") fea_dim = atoi(v.c_str()); else if (k == "fea_context") ctx = atoi(v.c_str());
        else if (k == "targ_offset") toff = atoi(v.c_str()); else if (k == "dropoutflag") dropoutflag = atoi(v.c_str());
        else if (k == "visible_omit") vis = (float)atof(v.c_str()); else if (k == "hid_omit") hid = (float)atof(v.c_str());
        else if (k == "bunchsize") bunch = atoi(v.c_str()); else if (k == "traincache") cache = atoi(v.c_str());
        else if (k == "activation") activation = v == "sigmoid" ? 1 : 0; else if (k == "device") device = atoi(v.c_str());
        else if (k == "compute") compute = v == "bf16" ? 1 : 0;
        else if (k == "layersizes") {
            size_t pos = 0;
            while (L < MAXLAYER) {
                const size_t c = v.find(',', pos);
                ls[L++] = atoi(v.substr(pos, c == std::string::npos ? c : c - pos).c_str());
                if (c == std::string::npos) break;
                pos = c + 1;
            }
        }
    }
    if (L < 2 || L > MAXLAYER - 1 || fea_dim < 1 || ctx < 1 || toff < 0 || toff >= ctx || cache < 1 || cache > MAXCACHEFRAME || bunch < 1) {
        printf("bpforward: need layersizes (2..%d sizes), fea_dim, fea_context, 0 <= targ_offset < fea_context, traincache <= %d\n", MAXLAYER - 1, MAXCACHEFRAME);
        exit(0);
    }
    const int sL = ls[L - 1];
    bp::ReaderConfig rc;
    rc.fea_file = fea_file; rc.targ_file = fea_file;          // no targets at enhancement time: the feature file stands in (unused)
    rc.norm_file = norm_file; rc.fea_dim = fea_dim; rc.fea_context = ctx; rc.targ_offset = toff; rc.out_dim = fea_dim;
    rc.traincache = cache; rc.input_dim = ls[0];
    bp::PfileReader reader(rc);
    reader.open();
    std::vector<std::vector<float>> Wv(L), Bv(L);
    float *weights[MAXLAYER] = {0}, *bias[MAXLAYER] = {0};
    for (int i = 1; i < L; ++i) { Wv[i].assign((size_t)ls[i] * ls[i - 1], 0.f); Bv[i].assign(ls[i], 0.f); weights[i] = Wv[i].data(); bias[i] = Bv[i].data(); }
    FILE *fi = fopen(wts_file.c_str(), "rb");
    if (!fi) { printf("can not open initial weights file: %s\n", wts_file.c_str()); exit(0); }
    const std::string err = bp::read_weights(fi, L, ls, weights, bias);
    fclose(fi);
    if (!err.empty()) { printf("%s\n", err.c_str()); exit(0); }
    int st = 0, en = (int)reader.total_sents() - 1;
    if (!range.empty()) { const size_t d = range.find('-'); if (d == std::string::npos) { printf("sent range: %s format error.\n", range.c_str()); exit(0); }
                          st = atoi(range.substr(0, d).c_str()); en = atoi(range.substr(d + 1).c_str()); }
    if (st < 0 || en >= (int)reader.total_sents() || st > en) { printf("sent range: %d to %d number error.\n", st, en); exit(0); }

    bp_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.gpu_used = 1; cfg.numlayers = L;
    for (int i = 0; i < L; ++i) cfg.layersizes[i] = ls[i];
    cfg.bunchsize = bunch; cfg.lrate = 0.f; cfg.momentum = 0.f; cfg.dropoutflag = dropoutflag; cfg.visible_omit = vis; cfg.hid_omit = hid;
    cfg.activation = activation; cfg.device = device; cfg.compute_dtype = compute; cfg.max_chunk_frames = cache;
    bp_handle *h = nullptr;
    if (bp_create(&cfg, weights, bias, &h) != 0) { printf("%s\n", bp_last_error()); exit(0); }

    // ---- output Pfile: header, records in reader order, sentence table
    FILE *fo = fopen(out_file.c_str(), "wb");
    if (!fo) { printf("can not open output file: %s\n", out_file.c_str()); exit(0); }
    std::vector<char> header(32768, 0);
    fwrite(header.data(), 1, header.size(), fo);                       // rewritten at the end with the counts
    const std::vector<int> &fbs = reader.frames_before_sent();         // end offset (frames) of every sentence
    const int nsent = en - st + 1;
    std::vector<uint32_t> per_sent(nsent, 0);
    const bp::PfileReader::Plan plan = reader.plan_inference(st, en);    // every window exactly once (no training-style cut losses)
    bp::PfileReader::WindowChunk w;
    std::vector<float> out;
    std::vector<uint32_t> rec(2 + sL);
    unsigned total = 0;
    for (int c = 0; c < (int)plan.chunk_frame_st.size(); ++c) {
        const int n = reader.read_chunk_windows(plan, c, false, w);
        if (n <= 0) continue;
        bp_window_chunk d;
        memset(&d, 0, sizeof(d));
        d.n_samples = w.n_samples; d.n_frames = w.n_frames; d.fea_dim = w.fea_dim; d.context = ctx; d.n_nat = w.n_nat();
        d.fea = w.fea.data(); d.nat = w.nat.empty() ? nullptr : w.nat.data(); d.win_start = w.win_start.data();
        d.nat_row = w.nat_row.empty() ? nullptr : w.nat_row.data();
        out.resize((size_t)n * sL);
        if (bp_forward_windows(h, &d, out.data()) != 0) { printf("%s\n", bp_last_error()); exit(0); }
        for (int i = 0; i < n; ++i) {
            const int gframe = plan.chunk_frame_st[c] + w.win_start[i];            // first frame of the window, file-global
            const int s = (int)(std::upper_bound(fbs.begin(), fbs.end(), gframe) - fbs.begin());
            const int s_begin = s == 0 ? 0 : fbs[s - 1];
            rec[0] = be32((uint32_t)(s - st)); rec[1] = be32((uint32_t)(gframe - s_begin + toff));
            for (int k = 0; k < sL; ++k) { uint32_t u; memcpy(&u, &out[(size_t)i * sL + k], 4); rec[2 + k] = be32(u); }
            fwrite(rec.data(), 4, rec.size(), fo);
            if (s - st >= 0 && s - st < nsent) per_sent[s - st]++;
            ++total;
        }
    }
    uint32_t cum = 0, v = be32(0);
    fwrite(&v, 4, 1, fo);
    for (int s = 0; s < nsent; ++s) { cum += per_sent[s]; v = be32(cum); fwrite(&v, 4, 1, fo); }
    snprintf(header.data(), header.size(), "-pfile_header version 0 size 32768\n-num_sentences %d\n-num_frames %u\n-first_feature_column 2\n-num_features %d\n-end\n",
             nsent, total, sL);
    fseek(fo, 0, SEEK_SET);
    fwrite(header.data(), 1, header.size(), fo);
    fclose(fo);
    bp_destroy(h);
    printf("bpforward: %u frames of %d sentences enhanced -> %s\n", total, nsent, out_file.c_str());
    return 1;
}
