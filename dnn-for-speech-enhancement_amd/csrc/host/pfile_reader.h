// pfile_reader.h -- host-side producer of the trainer's inputs ("next" row N1 of SURVEY.md 8f):
// ICSI Pfile pair (features + float targets) -> chunks of stacked, mean/variance-normalised input
// frames + targets, exactly what the reference's Interface feeds to BP_GPU::train / CrossValid
// (Interface.cc:468-1034, restated from its behaviour; formats in SURVEY.md Appendix B).
//
// Parity: pinned BYTE FOR BYTE to the reference's own Interface.cc, which compiles in the build container once its
// `#include "BP_GPU.h"` resolves to this repo's drop-in header (`make -C oracle ref`): reference-generated
// fixtures tests/golden/ref/*.npz + tests/test_ref_pins.py (plan, chunk order, every chunk's indata/targ, .wts bytes),
// and additionally by the independent numpy restatement tests/pfile_util.py (tests/test_pfile_reader.py).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

namespace bp {

struct ReaderConfig {
    std::string fea_file, targ_file, norm_file;
    int fea_dim = 0, fea_context = 1, targ_offset = 0, out_dim = 0;
    int traincache = 0;          // samples per chunk
    int input_dim = 0;           // layersizes[0]: fea_dim*ctx (plain) or fea_dim*(ctx+1) (NAT block appended)
};

class PfileReader {
public:
    explicit PfileReader(const ReaderConfig &cfg);
    ~PfileReader();
    // Interface::get_pfile_info (Interface.cc:468-555): headers, sentence tables, consistency checks
    void open();
    // worker threads of one frame conversion (0 = up to 8, the single-reader default); the node-level ring sets cores / ranks
    void set_convert_threads(int n) { convert_threads_ = n; }
    // Interface::get_chunk_info[_cv] (Interface.cc:558-686): plan chunks over sentences [st, en] (inclusive)
    struct Plan {
        std::vector<int> chunk_frame_st; int sent_st = 0, sent_en = 0; unsigned total_samples = 0;
        // plan_inference only: explicit end frame and window count of every chunk (chunks may overlap by ctx-1 frames)
        std::vector<int> chunk_frame_en, chunk_samples;
    };
    Plan plan(int sent_st, int sent_en) const;
    // Chunking for ENHANCEMENT (no reference counterpart: its decoder is external).  The training planner above cuts
    // wherever the cache fills and drops the ctx-1 windows that straddle the cut (Interface.cc:607-614) -- harmless
    // for SGD, wrong for a tool that must emit every frame.  Here chunks end on sentence boundaries; a sentence
    // with more windows than the cache is split into pieces that overlap by ctx-1 frames.  Every window of every
    // sentence is emitted exactly once, in file order; the noise-aware block of a later piece is still computed
    // from the SENTENCE's first frames.
    Plan plan_inference(int sent_st, int sent_en) const;
    // Interface::Readchunk / Readchunk_cv (Interface.cc:689-1034): returns the number of samples; rows are
    // written at a shuffled position when `shuffle` (train) else in order (CV).  in: [samples][input_dim],
    // targ: [samples][out_dim].
    int read_chunk(const Plan &p, int chunk_index, bool shuffle, float *in, float *targ);
    // The same chunk WITHOUT the host-side stacking ("next" row N3): raw normalised frames + raw target frames +
    // one NAT row per sentence segment + per-sample tables.  Sample i (already at its shuffled position) is
    //   in[i]   = fea[win_start[i] .. win_start[i]+ctx) ++ (nat ? nat[nat_row[i]] : {})
    //   targ[i] = targ[targ_frame[i]]
    // read_chunk() is expand(read_chunk_windows()), so both produce identical samples by construction; the
    // library expands the same tables on the device (bp_train_chunk_windows).  Returns the number of samples.
    struct WindowChunk {
        int n_samples = 0, n_frames = 0;
        std::vector<float> fea, targ, nat;                    // [n_frames][fea_dim], [n_frames][out_dim], [n_nat][fea_dim]
        std::vector<int> win_start, targ_frame, nat_row;      // [n_samples]
        int n_nat() const;
        int fea_dim = 0;
    };
    int read_chunk_windows(const Plan &p, int chunk_index, bool shuffle, WindowChunk &out);
    // The same without the print-and-exit convention, for callers that run on a helper thread (a read-ahead thread, the
    // shared ring's producer): the error text comes back (empty = fine) and the MAIN thread decides how to leave.
    std::string try_read_chunk_windows(const Plan &p, int chunk_index, bool shuffle, WindowChunk &out);
    // The same in pieces (read_chunk_windows is their composition), for a node-level shared reader (chunk_ring.h):
    // tables once, frame conversion in slices by whoever has cores to spare.  convert_frames / nat_rows use positioned
    // reads only and may run concurrently in several threads or forked processes.
    struct ChunkShape { int frame_st = 0, n_frames = 0, n_samples = 0; };
    ChunkShape chunk_shape(const Plan &p, int chunk_index) const;
    void convert_frames(const Plan &p, int chunk_index, int frame_st, int lo, int hi, float *fea, float *targ) const;
    std::string try_convert_frames(const Plan &p, int chunk_index, int frame_st, int lo, int hi, float *fea, float *targ) const;
    void build_tables(const Plan &p, int chunk_index, bool shuffle, int *win_start, int *targ_frame, int *nat_row,
                      std::vector<int> &seg_start, std::vector<int> &seg_sent);
    void nat_rows(const Plan &p, int chunk_index, const float *fea, const std::vector<int> &seg_start, const std::vector<int> &seg_sent, float *nat) const;
    std::string try_nat_rows(const Plan &p, int chunk_index, const float *fea, const std::vector<int> &seg_start, const std::vector<int> &seg_sent, float *nat) const;
    int fea_dim() const { return cfg_.fea_dim; }
    int out_dim() const { return cfg_.out_dim; }
    // host-side expansion of a window chunk into stacked rows (what Interface::Readchunk leaves in its buffers)
    void expand(const WindowChunk &w, float *in, float *targ) const;
    unsigned total_frames() const { return total_frames_; }
    unsigned total_sents() const { return total_sents_; }
    const std::vector<int> &frames_before_sent() const { return frames_before_sent_; }
    bool nat() const { return nat_; }
    // Interface::GetRandIndex (Interface.cc:1044-1055): back-to-front Fisher-Yates on lrand48()
    static void rand_index(int *vec, int len);

private:
    ReaderConfig cfg_;
    FILE *fp_data_ = nullptr, *fp_targ_ = nullptr;
    unsigned total_frames_ = 0, total_sents_ = 0;
    std::vector<int> frames_before_sent_;
    std::vector<float> mean_, dvar_;
    int convert_threads_ = 0;
    bool nat_ = false;
    bool clamp_warned_ = false;
};

[[noreturn]] void die(const char *fmt, ...);   // message + exit(0), the reference's error convention

}  // namespace bp
