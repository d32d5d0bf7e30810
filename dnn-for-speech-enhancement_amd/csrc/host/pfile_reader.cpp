#include "pfile_reader.h"

#include <algorithm>

#include <thread>

#include <errno.h>
#include <stdarg.h>
#include <string>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

namespace bp {

static const long PFILE_HEADER_SIZE = 32768;   // Interface.cc:13

void die(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vprintf(fmt, ap);
    va_end(ap);
    printf("\n");
    exit(0);                                    // reference convention (Interface.cc:246-265)
}

static inline uint32_t bswap(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24); }

static unsigned header_uint(const char *hdr, const char *key)     // Interface::get_uint, Interface.cc:1057-1075
{
    const char *p = strstr(hdr, key);
    if (!p) die("pfile header format is Not correct.");
    p += strlen(key);
    unsigned v = 0; int count = 0;
    sscanf(p, " %u%n", &v, &count);
    if (count <= 1) die("%s num in pfile header is Not correct.", key);
    return v;
}

// Interface::read_tail (Interface.cc:1077-1093): the table holds num_sentences+1 big-endian cumulative
// frame offsets; the first word is skipped, out[i] = frames before the END of sentence i.
static void read_tail(FILE *fp, long offset, unsigned nsent, std::vector<int> &out)
{
    out.resize(nsent);
    if (fseek(fp, offset + 4, SEEK_SET) != 0 || fread(out.data(), sizeof(int), nsent, fp) != nsent)
        die("pfile tail is Not correct.");
    for (unsigned i = 0; i < nsent; ++i) out[i] = (int)bswap((uint32_t)out[i]);
}

// The reference trusts the sentence table; every chunk's frame range is computed from it, so a corrupt one would size
// buffers and file offsets from garbage.  It must be non-decreasing, non-negative and end on the header's frame count.
static void check_tail(const std::vector<int> &t, unsigned total_frames, const char *what)
{
    int prev = 0;
    for (size_t i = 0; i < t.size(); ++i) {
        if (t[i] < prev || (unsigned)t[i] > total_frames) die("%s pfile tail is Not correct (sentence %zu ends at frame %d of %u).", what, i, t[i], total_frames);
        prev = t[i];
    }
    if (!t.empty() && (unsigned)t.back() != total_frames) die("%s pfile tail is Not correct (the last sentence ends at frame %d, the header promises %u).", what, t.back(), total_frames);
}

PfileReader::PfileReader(const ReaderConfig &cfg) : cfg_(cfg)
{
    if (cfg_.fea_dim < 1 || cfg_.fea_context < 1 || cfg_.out_dim < 1 || cfg_.traincache < 1 || cfg_.fea_dim > (1 << 20) || cfg_.fea_context > (1 << 12))
        die("fea_dim, fea_context, traincache and the layer sizes must be positive (fea_dim %d, fea_context %d, output %d, traincache %d).",
            cfg_.fea_dim, cfg_.fea_context, cfg_.out_dim, cfg_.traincache);
    // the reference demands layersizes[0] == fea_dim*ctx + fea_dim (NAT hard-wired, Interface.cc:395-399); the
    // original check fea_dim*ctx is kept as the NAT-off mode
    if (cfg_.input_dim == cfg_.fea_dim * (cfg_.fea_context + 1)) nat_ = true;
    else if (cfg_.input_dim == cfg_.fea_dim * cfg_.fea_context) nat_ = false;
    else die("feadim times (+ noise) context must be equal to layersizes[0]");
}

PfileReader::~PfileReader()
{
    if (fp_data_) fclose(fp_data_);
    if (fp_targ_) fclose(fp_targ_);
}

void PfileReader::open()
{
    if (!(fp_data_ = fopen(cfg_.fea_file.c_str(), "rb"))) die("can not open feature file: %s", cfg_.fea_file.c_str());
    if (!(fp_targ_ = fopen(cfg_.targ_file.c_str(), "rb"))) die("can not open target file: %s", cfg_.targ_file.c_str());
    // normalisation file: 1 header line, fea_dim means, 1 header line, fea_dim inverse std (Interface.cc:300-325)
    FILE *fn = fopen(cfg_.norm_file.c_str(), "rt");
    if (!fn) die("can not open normalization file: %s", cfg_.norm_file.c_str());
    char buff[1024];
    mean_.resize(cfg_.fea_dim); dvar_.resize(cfg_.fea_dim);
    if (!fgets(buff, sizeof(buff), fn)) die("normalization file too short");
    for (int j = 0; j < cfg_.fea_dim; ++j) { if (!fgets(buff, sizeof(buff), fn)) die("normalization file too short"); mean_[j] = (float)atof(buff); }
    if (!fgets(buff, sizeof(buff), fn)) die("normalization file too short");
    for (int j = 0; j < cfg_.fea_dim; ++j) { if (!fgets(buff, sizeof(buff), fn)) die("normalization file too short"); dvar_[j] = (float)atof(buff); }
    fclose(fn);

    std::vector<char> header(PFILE_HEADER_SIZE + 1, 0);
    if (fread(header.data(), PFILE_HEADER_SIZE, 1, fp_data_) != 1) die("Failed to read data pfile header.");
    total_sents_ = header_uint(header.data(), "-num_sentences");
    total_frames_ = header_uint(header.data(), "-num_frames");
    if (total_sents_ < 1 || total_sents_ > (1u << 28) || total_frames_ > (1u << 30)) die("pfile header: implausible sentence / frame count (%u / %u).", total_sents_, total_frames_);
    read_tail(fp_data_, (long)total_frames_ * (long)sizeof(float) * (2 + cfg_.fea_dim) + PFILE_HEADER_SIZE, total_sents_,
              frames_before_sent_);
    check_tail(frames_before_sent_, total_frames_, "data");
    if (fseek(fp_targ_, 0, SEEK_SET) != 0 || fread(header.data(), PFILE_HEADER_SIZE, 1, fp_targ_) != 1)
        die("Failed to read target pfile header.");
    const unsigned ts = header_uint(header.data(), "-num_sentences"), tf = header_uint(header.data(), "-num_frames");
    if (ts != total_sents_ || tf != total_frames_)
        die("frames or sentence num in target pfile and data pfile is not consistent.");
    std::vector<int> ttail;
    read_tail(fp_targ_, (long)tf * (long)sizeof(float) * (2 + cfg_.out_dim) + PFILE_HEADER_SIZE, ts, ttail);
    for (unsigned i = 0; i < total_sents_; ++i)
        if (ttail[i] != frames_before_sent_[i]) die("tails in target pfile and data pfile is not consistent---%u.", i);
}

PfileReader::Plan PfileReader::plan(int sent_st, int sent_en) const
{
    if (sent_en < sent_st || sent_st < 0 || sent_en >= (int)total_sents_)
        die("sent range: %d to %d number error.", sent_st, sent_en);
    Plan p; p.sent_st = sent_st; p.sent_en = sent_en;
    const int ctx = cfg_.fea_context, cache = cfg_.traincache;
    int cur_frame_id = sent_st == 0 ? 0 : frames_before_sent_[sent_st - 1];
    int cur_chunk_frames = 0;
    p.chunk_frame_st.push_back(cur_frame_id);
    for (int s = sent_st; s <= sent_en; ++s) {
        const int frames_inc = frames_before_sent_[s] - cur_frame_id;
        cur_frame_id = frames_before_sent_[s];
        const int lost = frames_inc >= ctx ? ctx - 1 : frames_inc;       // a sentence loses ctx-1 frames to stacking
        cur_chunk_frames += frames_inc - lost;
        while (cur_chunk_frames >= cache) {
            // the next chunk starts where the cache-th sample of this one ended; the samples that would
            // straddle the cut are lost (Interface.cc:607-614)
            const int next_st = cur_frame_id - (cur_chunk_frames - cache);
            if (next_st >= (int)total_frames_) { cur_chunk_frames = cache - 1; break; }   // (the reference would spin here)
            p.chunk_frame_st.push_back(next_st);
            cur_chunk_frames = (cur_frame_id - next_st > ctx - 1) ? (cur_frame_id - next_st - ctx + 1) : 0;
        }
    }
    p.total_samples = (unsigned)((p.chunk_frame_st.size() - 1) * (size_t)cache + cur_chunk_frames);
    return p;
}

PfileReader::Plan PfileReader::plan_inference(int sent_st, int sent_en) const
{
    if (sent_en < sent_st || sent_st < 0 || sent_en >= (int)total_sents_)
        die("sent range: %d to %d number error.", sent_st, sent_en);
    Plan p; p.sent_st = sent_st; p.sent_en = sent_en;
    const int ctx = cfg_.fea_context, cache = cfg_.traincache;
    int chunk_st = sent_st == 0 ? 0 : frames_before_sent_[sent_st - 1], chunk_samples = 0, pos = chunk_st;
    auto close_chunk = [&](int end_frame) {
        if (end_frame > chunk_st) { p.chunk_frame_st.push_back(chunk_st); p.chunk_frame_en.push_back(end_frame); p.chunk_samples.push_back(chunk_samples); p.total_samples += (unsigned)chunk_samples; }
        chunk_st = end_frame; chunk_samples = 0;
    };
    for (int s = sent_st; s <= sent_en; ++s) {
        const int s_end = frames_before_sent_[s], len = s_end - pos;
        int windows = len >= ctx ? len - ctx + 1 : 0;
        if (chunk_samples + windows > cache && chunk_samples > 0) close_chunk(pos);     // this sentence starts a new chunk
        int piece_st = pos;
        while (windows > cache) {                          // a sentence longer than the cache: pieces overlapping by ctx-1 frames
            chunk_st = piece_st; chunk_samples = cache;
            close_chunk(piece_st + cache + ctx - 1);
            piece_st += cache; windows -= cache;
        }
        if (piece_st != pos) chunk_st = piece_st;          // the rest of a split sentence starts the next chunk
        chunk_samples += windows;
        pos = s_end;
    }
    close_chunk(pos);
    if (p.chunk_frame_st.empty()) { p.chunk_frame_st.push_back(chunk_st); p.chunk_frame_en.push_back(chunk_st); p.chunk_samples.push_back(0); }
    return p;
}

void PfileReader::rand_index(int *vec, int len)
{
    for (int i = 0; i < len - 1; ++i) {
        const int idx = (int)(lrand48() % (len - i));
        const int tmp = vec[idx];
        vec[idx] = vec[len - 1 - i];
        vec[len - 1 - i] = tmp;
    }
}

// Byte-swap / normalisation of a chunk's records is per-frame independent: split the rows over a few threads
// (a 102400-frame chunk is 2 x 106 MB of records; one thread converts ~0.5 G floats/s, the GPU consumes a chunk
// in 90 ms).  Element-wise work only, so the result does not depend on the split.
// A worker never calls die() (= printf + exit(0)): two workers failing together would run exit() concurrently, which is
// undefined behaviour, and atexit handlers would run while the siblings still write into the shared slot.
// body(lo, hi) returns an error text (empty = fine); the first one is handed back to the CALLING thread after the join.
// max_threads > 0 caps the split: under the node-level ring every rank converts a slice at the same time, and N ranks x 8
// workers on the same cores only add scheduling noise (bptrain gives each rank cores / N of them).
template <class F>
static std::string parallel_rows(int n, int max_threads, F body)
{
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)(hw > 8 ? 8 : (hw < 1 ? 1 : hw));
    if (max_threads > 0 && nt > max_threads) nt = max_threads;
    if (n < 4096) nt = 1;
    if (nt == 1) return body(0, n);
    std::vector<std::thread> th;
    std::vector<std::string> err((size_t)nt);
    for (int t = 0; t < nt; ++t) {
        const int lo = (int)((long)n * t / nt), hi = (int)((long)n * (t + 1) / nt);
        th.emplace_back([=, &body, &err] { err[(size_t)t] = body(lo, hi); });
    }
    for (auto &t : th) t.join();
    for (auto &e : err) if (!e.empty()) return e;
    return std::string();
}

int PfileReader::WindowChunk::n_nat() const { return fea_dim > 0 ? (int)(nat.size() / (size_t)fea_dim) : 0; }

// ---- the chunk reader in pieces, so that a node-level shared reader (chunk_ring.h: bptrain gpu_used=N) can have
// rank 0 build the tables once and every rank convert a slice of the frames.  read_chunk_windows below is their
// composition, so both paths produce identical chunks by construction.
PfileReader::ChunkShape PfileReader::chunk_shape(const Plan &p, int ci) const
{
    ChunkShape c;
    const int nchunks = (int)p.chunk_frame_st.size();
    c.frame_st = p.chunk_frame_st[ci];
    if (!p.chunk_frame_en.empty()) {                       // plan_inference
        c.n_frames = p.chunk_frame_en[ci] - c.frame_st;
        c.n_samples = p.chunk_samples[ci];
    } else if (ci == nchunks - 1) {
        c.n_frames = frames_before_sent_[p.sent_en] - c.frame_st;
        c.n_samples = (int)p.total_samples - cfg_.traincache * ci;
    } else {
        c.n_frames = p.chunk_frame_st[ci + 1] - c.frame_st;
        c.n_samples = cfg_.traincache;
    }
    if (c.n_samples < 0) c.n_samples = 0;
    if (c.n_frames <= 0 || c.n_samples == 0) c.n_frames = 0;
    return c;
}

// positioned read of exactly `bytes`; an interrupted call is retried, a short file is an error text (empty = fine)
static std::string pread_all(FILE *fp, void *dst, size_t bytes, long off, const char *what, int ci)
{
    char *d = (char *)dst;
    while (bytes > 0) {
        const ssize_t n = pread(fileno(fp), d, bytes, off);
        if (n < 0 && errno == EINTR) continue;
        if (n <= 0) {
            char msg[160];
            snprintf(msg, sizeof(msg), "%s pfile: short read in chunk %d (%s).", what, ci, n == 0 ? "file ends before the frames its header promises" : strerror(errno));
            return msg;
        }
        d += n; off += n; bytes -= (size_t)n;
    }
    return std::string();
}

// frames [lo, hi) of the chunk that starts at file frame frame_st: big-endian records {sent_id, frame_id, feat[D]} ->
// mean/variance normalised features at fea + lo*D, raw targets (not normalised, Interface.cc:815-816) at targ + lo*OD.
// Positioned reads only (no shared file offset): several threads / forked processes may convert slices at once.
void PfileReader::convert_frames(const Plan &p, int ci, int frame_st, int lo, int hi, float *fea, float *targ) const
{
    const std::string err = try_convert_frames(p, ci, frame_st, lo, hi, fea, targ);
    if (!err.empty()) die("%s", err.c_str());                     // from the calling thread, after every worker has been joined
}

std::string PfileReader::try_convert_frames(const Plan &p, int ci, int frame_st, int lo, int hi, float *fea, float *targ) const
{
    const int D = cfg_.fea_dim, OD = cfg_.out_dim, n = hi - lo;
    if (n <= 0) return std::string();
    // every worker reads AND converts its own rows: the copy out of the page cache (2 x 105 MB for a 102400-frame chunk
    // of 257-bin frames) is as expensive as the arithmetic, and a single positioned read of the whole chunk made it the
    // serial part of the reader (1.0 M frames/s end to end against 1.16 M for the GPU alone, round 2)
    // Round 6: a worker walks its rows in BLOCKS of 256 frames through one small buffer that stays in its
    // core's cache between the positioned read and the conversion.  Before, every worker allocated its whole slice (13 MB per
    // worker and chunk: a fresh mapping, zero-filled, page-faulted, then filled by one pread and streamed through memory again by
    // the conversion) -- with 8 ranks converting at once the node-level ring delivered 3.9 M frames/s on 8 cores however the work
    // was split (profiles/r06_reader_ring_8ranks.txt); the arithmetic is unchanged, element for element.
    return parallel_rows(n, convert_threads_, [&](int a, int b) -> std::string {
        constexpr int BLK = 256;
        const int wmax = D > OD ? D : OD;
        std::vector<uint32_t> raw((size_t)BLK * (wmax + 2));
        const float *mean = mean_.data(), *dvar = dvar_.data();
        for (int i0 = a; i0 < b; i0 += BLK) {
            const int i1 = i0 + BLK < b ? i0 + BLK : b;
            std::string e = pread_all(fp_data_, raw.data(), (size_t)(i1 - i0) * (D + 2) * 4, PFILE_HEADER_SIZE + (long)(frame_st + lo + i0) * (long)sizeof(float) * (D + 2), "data", ci);
            if (!e.empty()) return e;
            if (lo + i0 == 0) {
                const int first_sent = (int)bswap(raw[0]);          // only the first record's sentence id is used (Interface.cc:740-741)
                // The id indexes the sentence table.  The reference trusts it; a corrupt or mismatched Pfile would make us read
                // outside the table (or silently build wrong windows), so it must lie in the planned range and own the chunk's first frame.
                if (first_sent < p.sent_st || first_sent > p.sent_en || first_sent >= (int)total_sents_ || frames_before_sent_[first_sent] <= frame_st ||
                    (first_sent > 0 && frames_before_sent_[first_sent - 1] > frame_st)) {
                    char msg[200];
                    snprintf(msg, sizeof(msg), "data pfile: record %d carries sentence id %d, which does not contain that frame (sentences %d-%d planned).", frame_st, first_sent, p.sent_st, p.sent_en);
                    return msg;
                }
            }
            for (int i = i0; i < i1; ++i) {
                const uint32_t *src = raw.data() + (size_t)(i - i0) * (D + 2) + 2;
                float *dst = fea + (size_t)(lo + i) * D;
                for (int j = 0; j < D; ++j) {
                    const uint32_t x = __builtin_bswap32(src[j]);
                    float v; memcpy(&v, &x, 4);
                    v -= mean[j];
                    v *= dvar[j];
                    dst[j] = v;
                }
            }
            if (!targ) continue;
            e = pread_all(fp_targ_, raw.data(), (size_t)(i1 - i0) * (OD + 2) * 4, PFILE_HEADER_SIZE + (long)(frame_st + lo + i0) * (long)sizeof(float) * (OD + 2), "targ", ci);
            if (!e.empty()) return e;
            for (int i = i0; i < i1; ++i) {
                const uint32_t *src = raw.data() + (size_t)(i - i0) * (OD + 2) + 2;
                uint32_t *dst = reinterpret_cast<uint32_t *>(targ + (size_t)(lo + i) * OD);
                for (int j = 0; j < OD; ++j) dst[j] = __builtin_bswap32(src[j]);
            }
        }
        return std::string();
    });
}

// per-sample tables of the chunk (consumes the lrand48 stream when shuffle): samples per sentence segment inside the
// chunk, ctx stacked frames (oldest first).  seg_start / seg_sent: first chunk-relative frame and sentence of every
// segment that owns at least one window (one noise-aware row each, in this order).
void PfileReader::build_tables(const Plan &p, int ci, bool shuffle, int *win_start, int *targ_frame, int *nat_row,
                               std::vector<int> &seg_start, std::vector<int> &seg_sent)
{
    const ChunkShape c = chunk_shape(p, ci);
    const int ctx = cfg_.fea_context, frames_need = c.n_frames, samples = c.n_samples, frame_st = c.frame_st;
    seg_start.clear(); seg_sent.clear();
    std::vector<int> sample_index(samples);
    for (int i = 0; i < samples; ++i) sample_index[i] = i;
    if (shuffle) rand_index(sample_index.data(), samples);
    for (int i = 0; i < samples; ++i) { win_start[i] = 0; targ_frame[i] = 0; if (nat_row) nat_row[i] = 0; }
    if (frames_need <= 0 || samples <= 0) return;
    // the sentence that owns the chunk's first frame (the reader cross-checks the record's own id in convert_frames)
    int cur_sent = (int)(std::upper_bound(frames_before_sent_.begin(), frames_before_sent_.end(), frame_st) - frames_before_sent_.begin());
    int frames_processed = 0, cur_frame_id = frame_st, cur_sample = 0;
    while (frames_processed != frames_need && cur_sent < (int)total_sents_) {
        int seg;
        if (frames_before_sent_[cur_sent] > frames_need + frame_st) seg = frames_need - frames_processed;
        else seg = frames_before_sent_[cur_sent] - cur_frame_id;
        int nat_id = -1;
        for (int j = 0; j <= seg - ctx && cur_sample < samples; ++j) {
            const int pos = sample_index[cur_sample];
            win_start[pos] = frames_processed + j;
            if (nat_row) {
                if (nat_id < 0) { nat_id = (int)seg_start.size(); seg_start.push_back(frames_processed); seg_sent.push_back(cur_sent); }
                nat_row[pos] = nat_id;
            }
            int tf = frames_processed + j + cfg_.targ_offset;
            if (tf >= frames_need) {                     // (the reference reads past its buffer here)
                if (!clamp_warned_) { fprintf(stderr, "pfile_reader: targ_offset %d points past the chunk's last frame; clamped (check targ_offset)\n", cfg_.targ_offset); clamp_warned_ = true; }
                tf = frames_need - 1;
            }
            targ_frame[pos] = tf;
            ++cur_sample;
        }
        cur_frame_id = frames_before_sent_[cur_sent];
        ++cur_sent;
        frames_processed += seg;
    }
    // samples the segment walk did not reach (cannot happen with a consistent plan) keep window 0: the stacked
    // reader would have left stale rows there
}

// noise-aware training: one row per segment = mean of the segment's first 6 normalised frames, summed left to right
// and divided by 6.0f (Interface.cc:776-779, generalised from the literal 129 to fea_dim)
void PfileReader::nat_rows(const Plan &p, int ci, const float *fea, const std::vector<int> &seg_start, const std::vector<int> &seg_sent, float *nat) const
{
    const std::string err = try_nat_rows(p, ci, fea, seg_start, seg_sent, nat);
    if (!err.empty()) die("%s", err.c_str());
}

std::string PfileReader::try_nat_rows(const Plan &p, int ci, const float *fea, const std::vector<int> &seg_start, const std::vector<int> &seg_sent, float *nat) const
{
    const ChunkShape c = chunk_shape(p, ci);
    const int D = cfg_.fea_dim, frames_need = c.n_frames;
    const bool inference = !p.chunk_frame_en.empty();
    for (size_t sg = 0; sg < seg_start.size(); ++sg) {
        float *nrow = nat + sg * (size_t)D;
        const int cur_sent = seg_sent[sg], sent_begin = cur_sent == 0 ? 0 : frames_before_sent_[cur_sent - 1];
        if (inference && c.frame_st + seg_start[sg] > sent_begin) {
            // a later piece of a sentence split by plan_inference: the noise estimate is still the mean of the
            // SENTENCE's first 6 frames, which lie before this chunk -- fetch and normalise just those
            const int nf = std::min(6, frames_before_sent_[cur_sent] - sent_begin);
            std::vector<uint32_t> head((size_t)nf * (D + 2));
            const std::string e = pread_all(fp_data_, head.data(), head.size() * 4, PFILE_HEADER_SIZE + (long)sent_begin * (long)sizeof(float) * (D + 2), "data", ci);
            if (!e.empty()) return e;
            for (int k = 0; k < D; ++k) {
                float sacc = 0.0f;
                for (int f = 0; f < 6; ++f) {
                    const uint32_t x = bswap(head[(size_t)(f < nf ? f : nf - 1) * (D + 2) + 2 + k]);
                    float v; memcpy(&v, &x, 4);
                    v -= mean_[k]; v *= dvar_[k];
                    sacc = f == 0 ? v : sacc + v;
                }
                nrow[k] = sacc / 6.0f;
            }
            continue;
        }
        for (int k = 0; k < D; ++k) {
            float s = 0.0f;
            for (int f = 0; f < 6; ++f) {
                int fr = seg_start[sg] + f;
                if (fr >= frames_need) fr = frames_need - 1;          // (the reference reads past its buffer here)
                s = f == 0 ? fea[(size_t)fr * D + k] : s + fea[(size_t)fr * D + k];
            }
            nrow[k] = s / 6.0f;
        }
    }
    return std::string();
}

int PfileReader::read_chunk_windows(const Plan &p, int ci, bool shuffle, WindowChunk &w)
{
    const std::string err = try_read_chunk_windows(p, ci, shuffle, w);
    if (!err.empty()) die("%s", err.c_str());
    return w.n_samples;
}

std::string PfileReader::try_read_chunk_windows(const Plan &p, int ci, bool shuffle, WindowChunk &w)
{
    const int D = cfg_.fea_dim, OD = cfg_.out_dim;
    const ChunkShape c = chunk_shape(p, ci);
    w.fea_dim = D;
    w.n_samples = c.n_samples;
    w.n_frames = c.n_frames;
    w.nat.clear();
    w.win_start.assign(w.n_samples, 0); w.targ_frame.assign(w.n_samples, 0); w.nat_row.assign(nat_ ? w.n_samples : 0, 0);
    std::vector<int> seg_start, seg_sent;
    // (tables first: the shuffle consumes lrand48 before any file access, as Readchunk does, Interface.cc:700-704)
    build_tables(p, ci, shuffle, w.win_start.data(), w.targ_frame.data(), nat_ ? w.nat_row.data() : nullptr, seg_start, seg_sent);
    if (c.n_frames <= 0) return std::string();
    w.fea.resize((size_t)c.n_frames * D);
    w.targ.resize((size_t)c.n_frames * OD);
    std::string err = try_convert_frames(p, ci, c.frame_st, 0, c.n_frames, w.fea.data(), w.targ.data());
    if (!err.empty()) return err;
    if (nat_) {
        w.nat.resize(seg_start.size() * (size_t)D);
        err = try_nat_rows(p, ci, w.fea.data(), seg_start, seg_sent, w.nat.data());
    }
    return err;
}

void PfileReader::expand(const WindowChunk &w, float *in, float *targ) const
{
    const int D = cfg_.fea_dim, ctx = cfg_.fea_context, OD = cfg_.out_dim, s0 = cfg_.input_dim;
    if (w.n_frames <= 0) return;
    for (int i = 0; i < w.n_samples; ++i) {
        float *row = in + (size_t)i * s0;
        memcpy(row, &w.fea[(size_t)w.win_start[i] * D], sizeof(float) * (size_t)ctx * D);
        if (nat_) memcpy(row + (size_t)ctx * D, &w.nat[(size_t)w.nat_row[i] * D], sizeof(float) * D);
        memcpy(targ + (size_t)i * OD, &w.targ[(size_t)w.targ_frame[i] * OD], sizeof(float) * OD);
    }
}

int PfileReader::read_chunk(const Plan &p, int ci, bool shuffle, float *in, float *targ)
{
    WindowChunk w;
    const int n = read_chunk_windows(p, ci, shuffle, w);
    expand(w, in, targ);
    return n;
}

}  // namespace bp
