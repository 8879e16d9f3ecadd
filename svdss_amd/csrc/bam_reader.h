// bam_reader.h -- BGZF/BAM record reader for the host CLI (zlib only; blocks inflated in parallel, one
// chunk ahead of the parser).
// Stands where htslib's hts_open / sam_hdr_read / sam_read1 / bam_aux_get stand in
// /root/reference/ping_pong.cpp:58,247-249,196-201.  Only what `search` consumes is
// decoded: flag, refID, l_seq, read name, 4-bit SEQ, integer aux tags (XF, HP).
// (The block inflaters are in bgzf_inflater.h.)
#pragma once
#include "bgzf_inflater.h"

struct BamRecord {
  int32_t tid = -1, pos = 0, l_seq = 0;
  uint16_t flag = 0;
  uint8_t mapq = 0;
  std::string qname;
  std::vector<uint8_t> seq4;   // packed 4-bit bases, (l_seq+1)/2 bytes
  std::vector<uint8_t> aux;    // raw aux block
  std::vector<uint32_t> cigar; // raw BAM cigar words (len << 4 | op)
  std::vector<uint8_t> qual;   // l_seq bytes
  uint16_t bin = 0;
  int32_t mtid = -1, mpos = -1, isize = 0;

  // seq_nt16_str decoding of the 4-bit bases
  std::string seq_string() const {
    static const char NT16[] = "=ACMGRSVTWYHKDBN";
    std::string s((size_t)l_seq, 'N');
    for (int i = 0; i < l_seq; ++i) s[(size_t)i] = NT16[(seq4[(size_t)i >> 1] >> ((~i & 1) << 2)) & 0xf];
    return s;
  }
  // bam_endpos: position after the last reference base covered
  int32_t endpos() const {
    int32_t r = 0;
    for (uint32_t c : cigar) {
      const uint32_t op = c & 0xf;
      if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) r += (int32_t)(c >> 4);
    }
    return pos + (r ? r : 1);
  }
};

class BamReader {
 public:
  explicit BamReader(const std::string& path, int threads = 0) : f_(fopen(path.c_str(), "rb")) {
    if (f_) {   // regular files are mapped: blocks are inflated straight from the page cache, no read() copies
      struct stat st;
      if (fstat(fileno(f_), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && !getenv("SVDSS_BAM_STREAM")) {
        // regular files: every chunk loader reads its own range with pread into a private (recycled) buffer, in
        // parallel with the others; the blocks are then located in ticket order.  (Inflating straight from a mapping
        // of the file was tried first: with a hundred workers the page faults on one mapping serialise in the kernel.)
        pread_size_ = (size_t)st.st_size;
      }
      if (getenv("SVDSS_BAM_MMAP") && fstat(fileno(f_), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        pread_size_ = 0;
        void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(f_), 0);
        if (m != MAP_FAILED) { map_ = (const uint8_t*)m; map_size_ = (size_t)st.st_size; (void)madvise(m, map_size_, MADV_SEQUENTIAL); }
      }
    }
    const unsigned hw = effective_cpus();
    threads_ = threads > 0 ? threads : (int)std::max(1u, std::min(128u, hw > 32 ? hw / 2 : hw));
    pool_.reset(new InflatePool(threads_));
    if (const char* e = getenv("SVDSS_BAM_AHEAD")) ahead_ = (size_t)std::max(1, atoi(e));
    if (const char* e = getenv("SVDSS_BAM_SLAB_KB")) slab_ = (size_t)std::max(64, atoi(e)) << 10;
    // the page tables of the mapping are filled ahead of the block scanner and the inflate workers, 16 MB per call:
    // without it every BGZF block costs its worker a page fault, and faults of many threads on one mapping serialise
    // in the kernel (the inflate rate did not move between 48 and 224 workers)
    if (map_ && getenv("SVDSS_BAM_POPULATE")) prefault_ = std::thread([this] { prefault_loop(); });
  }
  // SVDSS_DEBUG: the reader's counters on stderr (once; also called by a program that ends without destructors)
  void report() {
    if (reported_ || !getenv("SVDSS_DEBUG")) return;
    reported_ = true;
    report_impl();
  }
  ~BamReader() {
    { std::lock_guard<std::mutex> lk(file_m_); prefault_stop_ = true; }
    file_cv_.notify_all();
    if (prefault_.joinable()) prefault_.join();
    report();
    drain();
    pool_.reset();
    for (GpuObj& g : gpu_objs_) { if (g.h) gpu_.inflate_free(g.h); if (g.d_out) gpu_.device_free(g.dev, g.d_out); }
    if (map_) munmap((void*)map_, map_size_);
    if (f_) fclose(f_);
  }
  BamReader(const BamReader&) = delete;
  BamReader& operator=(const BamReader&) = delete;
  bool ok() const { return f_ != nullptr; }
  // chunks located / inflated ahead of the parser (before the first read; a reader that only looks at the first few
  // thousand records should not pull the whole file through the inflater)
  void set_ahead(size_t n) { ahead_ = std::max<size_t>(1, n); }

  // BGZF blocks inflated on the GPU (csrc/inflate.hip) instead of by the host workers: `percent` of the chunks (the
  // rest stay with the host pool, which otherwise idles); the CRC32 of every block is still checked here, on the host.
  // The entry points come as function pointers so that this header stays free of the library's.
  struct GpuInflateApi {
    int (*inflate)(void** obj, int device, const uint8_t* comp, int64_t comp_bytes, const void* blocks, int64_t n_blocks,
                   void* d_out, uint8_t* host_out, int64_t out_bytes, int64_t* bad_block) = nullptr;
    void (*inflate_free)(void* obj) = nullptr;
    int (*device_alloc)(int device, int64_t bytes, void** out) = nullptr;
    void (*device_free)(int device, void* p) = nullptr;
    int (*host_alloc)(int64_t bytes, void** out) = nullptr;
    void (*host_free)(void* p) = nullptr;
  };
  // (n_devices > 1: the chunks go to devices device, device + 1, ... in turn -- `search --gpus N` inflates on all of them)
  void enable_gpu_inflate(const GpuInflateApi& api, int device, int percent, int n_devices = 1) {
    gpu_ = api; gpu_device_ = device; gpu_percent_ = std::max(0, std::min(101, percent));   // (101: every chunk)
    gpu_n_devices_ = std::max(1, n_devices);
    pin_hooks().alloc = api.host_alloc; pin_hooks().free_ = api.host_free;
    // A page-locked allocation of a chunk's size takes ~0.2 s: it pays when the buffer is used again and again (a
    // 15 GB file: thirty times per loader), not when the whole file is a handful of chunks -- those go through ordinary
    // memory (a pageable copy of 200 MB costs a tenth of that).  SVDSS_PIN_MIN_CHUNKS overrides the threshold.
    {
      const char* e = getenv("SVDSS_PIN_MIN_CHUNKS");
      const size_t min_chunks = e && *e ? (size_t)atoll(e) : 4 * ahead_;
      pin_buffers_ = pread_size_ == 0 || (pread_size_ + slab_ - 1) / slab_ >= min_chunks;
    }
  }
  // Page-locked chunk buffers ahead of the first read.  A page-locked allocation of a chunk's size takes ~0.1 s, and the
  // first `ahead` loaders would each pay for two of them before the first record is seen: a caller with something else
  // to do first (`search` restores its index) runs this beside it.  est_ratio: inflated / compressed size expected.
  void prewarm(double est_ratio = 1.75) {
    if (!gpu_.inflate || !pread_size_ || !pin_hooks().alloc || !pin_buffers_) return;
    // compressed: one per loader in flight; inflated: those plus the chunks a batch of records keeps alive while its
    // packed bases are copied out (about as many again)
    const size_t n_file = (pread_size_ + slab_ - 1) / slab_;
    const size_t n_comp = std::min<size_t>(ahead_ + 2, n_file), n_out = std::min<size_t>(2 * ahead_, n_file);
    const size_t out_bytes = (size_t)((double)slab_ * est_ratio) + ((size_t)8 << 20);
    std::vector<std::thread> th;
    for (size_t t = 0; t < 8; ++t)
      th.emplace_back([this, t, n_comp, n_out, out_bytes] {
        for (size_t i = t; i < n_comp + n_out; i += 8) {
          try {
            Bytes b;
            b.alloc(i < n_comp ? slab_ + kOverlap : out_bytes, true);
            if (b.pin_refused) return;   // no more page-locked memory to be had: the loaders use ordinary buffers
            FreeList& fl = i < n_comp ? *comp_free_ : *free_;
            std::lock_guard<std::mutex> lk(fl.m);
            fl.v.push_back(std::move(b));
          } catch (const std::bad_alloc&) { return; }   // (the loaders allocate what they need, or fall back)
        }
      });
    for (std::thread& x : th) x.join();
  }
  const std::string& error() const { return err_; }
  const std::vector<std::string>& ref_names() const { return refs_; }
  const std::vector<int32_t>& ref_lens() const { return ref_lens_; }
  const std::string& header_text() const { return text_; }

  bool read_header() {
    char magic[4];
    if (!read(magic, 4) || memcmp(magic, "BAM\1", 4) != 0) { if (err_.empty()) err_ = "not a BAM file"; return false; }
    int32_t l_text, n_ref;
    if (!read(&l_text, 4)) return fail("truncated header");
    if (l_text < 0) return fail("corrupt header");
    // (the text and the names are read in pieces: a length field is not trusted with an allocation of its size)
    std::string text;
    if (!read_string(text, (size_t)l_text)) return fail("truncated header");
    text_ = text;
    if (!read(&n_ref, 4)) return fail("truncated header");
    if (n_ref < 0) return fail("corrupt header");
    for (int i = 0; i < n_ref; ++i) {
      int32_t l_name, l_ref;
      if (!read(&l_name, 4)) return fail("truncated header");
      if (l_name < 0) return fail("corrupt header");
      std::string name;
      if (!read_string(name, (size_t)l_name) || !read(&l_ref, 4)) return fail("truncated header");
      if (!name.empty() && name.back() == '\0') name.pop_back();
      refs_.push_back(name);
      ref_lens_.push_back(l_ref);
    }
    return true;
  }

  // 1 = record read, 0 = clean end of file, -1 = error
  // (want_qual = false leaves r.qual empty: `search` never looks at qualities)
  int next(BamRecord& r, bool want_qual = true) {
    int32_t block_size;
    size_t got = read_some(&block_size, 4);
    if (got == 0) return err_.empty() ? 0 : -1;
    if (got != 4 || block_size < 32) { err_ = "truncated record"; return -1; }
    buf_.resize((size_t)block_size);
    if (!read(buf_.data(), (size_t)block_size)) { err_ = "truncated record"; return -1; }
    const uint8_t* p = buf_.data();
    int32_t refID, pos, l_seq;
    uint8_t l_read_name, mapq;
    uint16_t n_cigar, flag;
    memcpy(&refID, p, 4);
    memcpy(&pos, p + 4, 4);
    l_read_name = p[8];
    mapq = p[9];
    memcpy(&n_cigar, p + 12, 2);
    memcpy(&flag, p + 14, 2);
    memcpy(&l_seq, p + 16, 4);
    size_t o = 32;
    if (l_seq < 0 || o + l_read_name + 4u * n_cigar + ((size_t)l_seq + 1) / 2 + (size_t)l_seq > (size_t)block_size) {
      err_ = "corrupt record";
      return -1;
    }
    r.tid = refID; r.pos = pos; r.l_seq = l_seq; r.flag = flag; r.mapq = mapq;
    r.qname.assign((const char*)p + o, l_read_name ? l_read_name - 1 : 0);
    o += l_read_name;
    r.cigar.resize(n_cigar);
    if (n_cigar) memcpy(r.cigar.data(), p + o, 4u * n_cigar);
    o += 4u * n_cigar;
    memcpy(&r.bin, p + 10, 2);
    memcpy(&r.mtid, p + 20, 4);
    memcpy(&r.mpos, p + 24, 4);
    memcpy(&r.isize, p + 28, 4);
    r.seq4.assign(p + o, p + o + ((size_t)l_seq + 1) / 2);
    o += ((size_t)l_seq + 1) / 2;
    if (want_qual) r.qual.assign(p + o, p + o + (size_t)l_seq); else r.qual.clear();
    o += (size_t)l_seq;
    r.aux.assign(p + o, p + block_size);
    return 1;
  }

  // page-locked buffers for the GPU inflate path (set by enable_gpu_inflate: the reader itself does not know HIP)
  struct PinHooks {
    int (*alloc)(int64_t, void**) = nullptr;
    void (*free_)(void*) = nullptr;
  };
  static PinHooks& pin_hooks() { static PinHooks h; return h; }

  // Page-locked memory is a bounded resource: the buffers in flight between the loaders and the parser are pinned, what
  // a caller keeps for long (the record cache of `call` pass 1) must not be.  Pinned bytes are counted and capped
  // (SVDSS_PIN_CAP_GB, default 8); past the cap, or when the runtime refuses, a buffer is ordinary memory -- the GPU
  // inflate path takes either kind.
  static std::atomic<long long>& pinned_bytes() { static std::atomic<long long> v{0}; return v; }
  static long long pin_cap_bytes() {
    static const long long cap = [] {
      const char* e = getenv("SVDSS_PIN_CAP_GB");
      const double gb = e && *e ? atof(e) : 8.0;
      return (long long)(gb * (double)(1ll << 30));
    }();
    return cap;
  }

  struct Bytes {   // uninitialised buffer (std::vector would zero-fill what inflate overwrites anyway)
    struct Deleter {
      bool pinned;
      size_t bytes;
      Deleter() : pinned(false), bytes(0) {}
      explicit Deleter(bool p_, size_t b_ = 0) : pinned(p_), bytes(b_) {}
      void operator()(uint8_t* q) const {
        if (pinned) { pin_hooks().free_(q); pinned_bytes().fetch_sub((long long)bytes); }
        else free(q);
      }
    };
    std::unique_ptr<uint8_t[], Deleter> p;
    size_t n = 0, cap = 0;
    bool pin_refused = false;   // this buffer asked for page-locked memory and got ordinary memory: do not ask again
    // (chunk-sized buffers: 2 MB-aligned and advised for huge pages -- 25 faults per 50 MB chunk instead of 12,800)
    void alloc(size_t k, bool pinned = false) {
      if (k > cap || !p || (pinned && !p.get_deleter().pinned && !pin_refused)) {
        size_t c = k ? k : 1;
        bool done = false;
        if (pinned && pin_hooks().alloc) {
          const size_t cp = (c + c / 8 + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
          void* q = nullptr;
          if (pinned_bytes().fetch_add((long long)cp) + (long long)cp <= pin_cap_bytes() &&
              pin_hooks().alloc((int64_t)cp, &q) == 0 && q) {
            p = std::unique_ptr<uint8_t[], Deleter>((uint8_t*)q, Deleter(true, cp));
            c = cp;
            done = true;
            pin_refused = false;
          } else {
            pinned_bytes().fetch_sub((long long)cp);
            pin_refused = true;          // cap reached or the runtime said no: pageable memory below
          }
        }
        if (done) {
        } else if (c >= ((size_t)4 << 20)) {
          c = (c + c / 8 + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
          void* q = nullptr;
          if (posix_memalign(&q, (size_t)2 << 20, c) != 0) throw std::bad_alloc();
          (void)madvise(q, c, MADV_HUGEPAGE);
          p = std::unique_ptr<uint8_t[], Deleter>((uint8_t*)q, Deleter(false));
        } else {
          p = std::unique_ptr<uint8_t[], Deleter>((uint8_t*)malloc(c), Deleter(false));
          if (!p) throw std::bad_alloc();
        }
        cap = c;
      }
      n = k;
    }
    uint8_t* data() { return p.get(); }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    void swap(Bytes& o) { p.swap(o.p); std::swap(n, o.n); std::swap(cap, o.cap); std::swap(pin_refused, o.pin_refused); }
  };

  struct FreeList { std::mutex m; std::vector<Bytes> v; };

  // A record sliced but not decoded: core fields + where name / cigar / packed bases / aux tags sit in an
  // arena the caller owns (qualities are skipped).  Slicing is sequential and runs at memcpy speed; the
  // caller decodes many records in parallel.
  struct RawRec {
    size_t off = 0;                 // arena offset of: name (l_name bytes, NUL-terminated), cigar, seq4, aux
    uint32_t l_name = 0, n_cigar = 0, l_aux = 0;
    int32_t tid = -1, pos = 0, l_seq = 0;
    uint16_t flag = 0;
    uint8_t mapq = 0;
    size_t name_off() const { return off; }
    size_t seq_off() const { return off + l_name + 4u * n_cigar; }
    size_t aux_off() const { return seq_off() + ((size_t)l_seq + 1) / 2; }
  };

  // growable byte buffer without the zero-fill of std::vector::resize (the arena is written exactly once)
  struct Arena {
    uint8_t* p = nullptr;
    size_t size = 0, cap = 0;
    Arena() = default;
    Arena(const Arena&) = delete;
    Arena& operator=(const Arena&) = delete;
    ~Arena() { free(p); }
    uint8_t* data() { return p; }
    const uint8_t* data() const { return p; }
    void clear() { size = 0; }
    void resize(size_t n) {
      if (n > cap) {
        size_t c = cap ? cap : (1u << 20);
        while (c < n) c += c / 2;
        uint8_t* q = (uint8_t*)realloc(p, c);
        if (!q) throw std::bad_alloc();
        p = q;
        cap = c;
      }
      size = n;
    }
  };

  // 1 = record appended to arena, 0 = clean end of file, -1 = error
  int next_raw(Arena& arena, RawRec& rr) {
    int32_t block_size;
    const size_t got = read_some(&block_size, 4);
    if (got == 0) return err_.empty() ? 0 : -1;
    uint8_t core[32];
    if (got != 4 || block_size < 32 || !read(core, 32)) { err_ = "truncated record"; return -1; }
    uint16_t n_cigar;
    memcpy(&rr.tid, core, 4);
    memcpy(&rr.pos, core + 4, 4);
    rr.l_name = core[8];
    rr.mapq = core[9];
    memcpy(&n_cigar, core + 12, 2);
    rr.n_cigar = n_cigar;
    memcpy(&rr.flag, core + 14, 2);
    memcpy(&rr.l_seq, core + 16, 4);
    if (rr.l_seq < 0) { err_ = "corrupt record"; return -1; }
    const size_t head = (size_t)rr.l_name + 4u * rr.n_cigar + ((size_t)rr.l_seq + 1) / 2;
    if (32 + head + (size_t)rr.l_seq > (size_t)block_size) { err_ = "corrupt record"; return -1; }
    rr.l_aux = (uint32_t)((size_t)block_size - 32 - head - (size_t)rr.l_seq);
    rr.off = arena.size;
    arena.resize(rr.off + head + rr.l_aux);
    if (!read(arena.data() + rr.off, head) || !skip((size_t)rr.l_seq) || !read(arena.data() + rr.off + head, rr.l_aux)) {
      err_ = "truncated record";
      return -1;
    }
    return 1;
  }

  // Zero-copy variant of next_raw: the record body stays where it was inflated.  `p` points at the 32-byte core block
  // (inside the current chunk, or inside `own` when the record straddles two chunks); the caller keeps the buffers
  // alive by holding chunk() (whenever chunk_id() changes) and `own`.
  struct RawView {
    const uint8_t* p = nullptr;
    std::shared_ptr<uint8_t> own;   // only for straddling records
    uint32_t l_name = 0, n_cigar = 0, l_aux = 0;
    int32_t tid = -1, pos = 0, l_seq = 0;
    uint16_t flag = 0;
    uint8_t mapq = 0;
    bool noqual = false;            // a slim record (svdss_bam_store_select): the tags follow the bases
    const uint8_t* name() const { return p + 32; }
    const uint8_t* seq4() const { return p + 32 + l_name + 4u * n_cigar; }
    const uint8_t* aux() const { return seq4() + ((size_t)l_seq + 1) / 2 + (noqual ? 0 : (size_t)l_seq); }
  };
  std::shared_ptr<Bytes> chunk() const { return chunk_; }
  uint64_t chunk_id() const { return chunk_id_; }

  // 1 = record, 0 = clean end of file, -1 = error
  int next_view(RawView& v) {
    int32_t block_size;
    const size_t got = read_some(&block_size, 4);
    if (got == 0) return err_.empty() ? 0 : -1;
    if (got != 4 || block_size < 32) { err_ = "truncated record"; return -1; }
    v.own.reset();
    v.noqual = false;
    if (chunk_->size() - upos_ >= (size_t)block_size) {   // whole record inside the current chunk
      v.p = chunk_->data() + upos_;
      upos_ += (size_t)block_size;
    } else {
      v.own = std::shared_ptr<uint8_t>(new uint8_t[(size_t)block_size], std::default_delete<uint8_t[]>());
      if (!read(v.own.get(), (size_t)block_size)) { err_ = "truncated record"; return -1; }
      v.p = v.own.get();
    }
    const uint8_t* core = v.p;
    uint16_t n_cigar;
    memcpy(&v.tid, core, 4);
    memcpy(&v.pos, core + 4, 4);
    v.l_name = core[8];
    v.mapq = core[9];
    memcpy(&n_cigar, core + 12, 2);
    v.n_cigar = n_cigar;
    memcpy(&v.flag, core + 14, 2);
    memcpy(&v.l_seq, core + 16, 4);
    if (v.l_seq < 0) { err_ = "corrupt record"; return -1; }
    const size_t head = 32 + (size_t)v.l_name + 4u * v.n_cigar + ((size_t)v.l_seq + 1) / 2 + (size_t)v.l_seq;
    if (head > (size_t)block_size) { err_ = "corrupt record"; return -1; }
    v.l_aux = (uint32_t)((size_t)block_size - head);
    return 1;
  }

  // a sliced record turned into a BamRecord (without qualities; the packed bases only if want_seq)
  static void materialize(const Arena& arena, const RawRec& rr, BamRecord& r, bool want_seq = true) {
    const uint8_t* p = arena.data() + rr.off;
    r.tid = rr.tid; r.pos = rr.pos; r.l_seq = rr.l_seq; r.flag = rr.flag; r.mapq = rr.mapq;
    r.qname.assign((const char*)p, rr.l_name ? rr.l_name - 1 : 0);
    r.cigar.resize(rr.n_cigar);
    if (rr.n_cigar) memcpy(r.cigar.data(), p + rr.l_name, 4u * rr.n_cigar);
    const uint8_t* sq = arena.data() + rr.seq_off();
    if (want_seq) r.seq4.assign(sq, sq + ((size_t)rr.l_seq + 1) / 2); else r.seq4.clear();
    r.qual.clear();
    const uint8_t* ax = arena.data() + rr.aux_off();
    r.aux.assign(ax, ax + rr.l_aux);
  }
  // the same from a zero-copy view
  static void materialize(const RawView& v, BamRecord& r, bool want_seq = true) {
    r.tid = v.tid; r.pos = v.pos; r.l_seq = v.l_seq; r.flag = v.flag; r.mapq = v.mapq;
    r.qname.assign((const char*)v.name(), v.l_name ? v.l_name - 1 : 0);
    r.cigar.resize(v.n_cigar);
    if (v.n_cigar) memcpy(r.cigar.data(), v.name() + v.l_name, 4u * v.n_cigar);
    if (want_seq) r.seq4.assign(v.seq4(), v.seq4() + ((size_t)v.l_seq + 1) / 2); else r.seq4.clear();
    r.qual.clear();
    r.aux.assign(v.aux(), v.aux() + v.l_aux);
  }
  static void materialize_seq(const Arena& arena, const RawRec& rr, BamRecord& r) {
    const uint8_t* sq = arena.data() + rr.seq_off();
    r.seq4.assign(sq, sq + ((size_t)rr.l_seq + 1) / 2);
  }

  // integer aux tag (bam_aux_get + bam_aux2i); returns false if absent or not an integer
  static bool aux_int(const BamRecord& r, const char tag[2], int64_t& out) {
    return aux_int(r.aux.data(), r.aux.size(), tag, out);
  }
  static bool aux_int(const uint8_t* p, size_t n, const char tag[2], int64_t& out) {
    const uint8_t* e = p + n;
    while (p + 3 <= e) {
      const char t0 = (char)p[0], t1 = (char)p[1], ty = (char)p[2];
      p += 3;
      size_t sz = 0;
      switch (ty) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'Z': case 'H': { const uint8_t* z = p; while (z < e && *z) ++z; sz = (size_t)(z - p) + 1; break; }
        case 'B': {
          if (p + 5 > e) return false;
          const char st = (char)p[0];
          int32_t cnt; memcpy(&cnt, p + 1, 4);
          const size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
          sz = 5 + es * (size_t)cnt;
          break;
        }
        default: return false;
      }
      if (p + sz > e) return false;
      if (t0 == tag[0] && t1 == tag[1]) {
        switch (ty) {
          case 'c': out = (int8_t)p[0]; return true;
          case 'C': out = p[0]; return true;
          case 's': { int16_t v; memcpy(&v, p, 2); out = v; return true; }
          case 'S': { uint16_t v; memcpy(&v, p, 2); out = v; return true; }
          case 'i': { int32_t v; memcpy(&v, p, 4); out = v; return true; }
          case 'I': { uint32_t v; memcpy(&v, p, 4); out = v; return true; }
          default: return false;
        }
      }
      p += sz;
    }
    return false;
  }

 private:
  bool fail(const char* m) { err_ = m; return false; }
  bool read(void* dst, size_t n) { return read_some(dst, n) == n; }
  // n bytes into a string that grows as they arrive (a damaged length field costs a failed read, not an allocation)
  bool read_string(std::string& out, size_t n) {
    out.clear();
    char piece[65536];
    while (n) {
      const size_t take = n < sizeof piece ? n : sizeof piece;
      if (!read(piece, take)) return false;
      out.append(piece, take);
      n -= take;
    }
    return true;
  }

  bool skip(size_t n) {
    while (n) {
      if (upos_ == chunk_->size()) {
        if (!next_chunk()) return false;
        continue;
      }
      const size_t take = std::min(n, chunk_->size() - upos_);
      upos_ += take;
      n -= take;
    }
    return true;
  }

  // reads up to n uncompressed bytes, refilling from inflated chunks of BGZF blocks
  size_t read_some(void* dst, size_t n) {
    size_t done = 0;
    while (done < n) {
      if (upos_ == chunk_->size()) {
        if (!next_chunk()) break;
        if (chunk_->empty()) continue;   // only empty blocks (EOF markers) -- keep going
      }
      const size_t take = std::min(n - done, chunk_->size() - upos_);
      memcpy((uint8_t*)dst + done, chunk_->data() + upos_, take);
      upos_ += take;
      done += take;
    }
    return done;
  }

  // BGZF blocks are independent deflate streams (<= 64 KiB each): a chunk of ~512 blocks (one 32 MB read) is located
  // sequentially, inflated by `threads_` workers, and the next chunk is prepared in the background while the
  // caller parses the current one (htslib's hts_set_threads plays this role for the reference).
  struct Chunk {
    Bytes data;
    bool eof = false;
    std::string err;
    std::string late_err;   // reported after the chunk's data has been consumed
  };
  struct BlockRef { size_t coff, clen, uoff; uint32_t isize, crc; };
  size_t slab_ = (size_t)32 << 20;   // compressed bytes read per chunk (~512 blocks; SVDSS_BAM_SLAB_KB: tests)

  bool next_chunk() {
    if (eof_seen_) {               // the final chunk was already handed out
      if (!pending_err_.empty()) { err_ = pending_err_; pending_err_.clear(); }
      return false;
    }
    while (pending_.size() < ahead_ && !launched_eof_) {
      const uint64_t ticket = n_launched_++;
      pending_.push_back(std::async(std::launch::async, [this, ticket] { return load_chunk(ticket); }));
    }
    if (pending_.empty()) { eof_seen_ = true; return false; }
    const auto tw0 = std::chrono::steady_clock::now();
    Chunk c = pending_.front().get();
    t_wait_ += (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tw0).count();
    pending_.pop_front();
    if (!c.err.empty()) { err_ = c.err; eof_seen_ = true; drain(); return false; }
    if (c.eof) { eof_seen_ = true; drain(); }
    if (!c.late_err.empty()) pending_err_ = c.late_err;
    // the buffer goes back to the free list when the last holder of the chunk lets go of it (a fresh 50-100 MB
    // allocation per chunk is a page fault per 4 KB for the inflate workers)
    {
      std::shared_ptr<FreeList> fl = free_;
      chunk_ = std::shared_ptr<Bytes>(new Bytes(), [fl](Bytes* b) {
        { std::lock_guard<std::mutex> lk(fl->m); if (fl->v.size() < 512) fl->v.push_back(std::move(*b)); }
        delete b;
      });
    }
    chunk_->swap(c.data);
    ++chunk_id_;
    upos_ = 0;
    if (c.eof && chunk_->empty()) {
      if (!pending_err_.empty()) { err_ = pending_err_; pending_err_.clear(); }
      return false;
    }
    return true;
  }

  void drain() {
    for (auto& f : pending_) if (f.valid()) f.wait();
    pending_.clear();
  }

  // runs on background threads.  File reads happen in ticket order (one loader at a time); the inflate of a
  // chunk overlaps the file read of the next one.
  Chunk load_chunk(uint64_t ticket) {
    // (std::async would carry an exception to the parser's get() and end the process: report it like any other error)
    try { return load_chunk_impl(ticket); }
    catch (const std::bad_alloc&) { Chunk c; c.err = "out of memory while loading a BAM chunk"; return c; }
    catch (const std::exception& e) { Chunk c; c.err = std::string("BAM chunk loader: ") + e.what(); return c; }
  }
  Chunk load_chunk_impl(uint64_t ticket) {
    Chunk c;
    std::vector<BlockRef> blocks;
    size_t total = 0;
    // pread mode: this loader's nominal range [base, base + slab) plus what the last block may need beyond it
    std::shared_ptr<Bytes> own;
    size_t own_got = 0;
    const size_t base = (size_t)ticket * slab_;
    // (nothing may leave this function before the ticket below is taken and released: the other loaders wait for it)
    const char* early_err = nullptr;
    if (pread_size_ && base < pread_size_) {
      try {
        own = take_comp();
        const size_t want = std::min(slab_ + kOverlap, pread_size_ - base);
        own->alloc(slab_ + kOverlap, gpu_.inflate != nullptr && pin_buffers_);
        while (own_got < want) {
          const ssize_t k = pread(fileno(f_), own->data() + own_got, want - own_got, (off_t)(base + own_got));
          if (k <= 0) break;
          own_got += (size_t)k;
        }
      } catch (const std::bad_alloc&) { early_err = "out of memory while loading a BAM chunk"; }
    }
    std::unique_lock<std::mutex> file_lock(file_m_);
    file_cv_.wait(file_lock, [&] { return next_ticket_ == ticket; });
    struct Release {   // hand the file to the next loader on every exit path
      BamReader* r; std::unique_lock<std::mutex>* lk; bool done = false;
      void operator()() { if (!done) { done = true; ++r->next_ticket_; lk->unlock(); r->file_cv_.notify_all(); } }
      ~Release() { (*this)(); }
    } release{this, &file_lock};
    if (early_err) { c.err = early_err; file_eof_ = true; return c; }
    if (file_eof_) { c.eof = true; return c; }
    const auto ts0 = std::chrono::steady_clock::now();
    // mapped file: the blocks of this chunk are located in the mapping; otherwise one large read per chunk (plus the
    // partial block the previous chunk left over)
    Bytes comp;
    const uint8_t* src;
    size_t avail, got;
    size_t start = 0;
    if (pread_size_) {
      if (base >= pread_size_ || next_off_ >= pread_size_) { c.eof = true; file_eof_ = true; return c; }
      if (next_off_ < base || next_off_ > base + slab_ + kOverlap) { c.err = "BGZF block chain lost"; file_eof_ = true; return c; }
      src = own->data();
      start = next_off_ - base;              // the first block of this chunk (the previous chunk's last one ended here)
      got = std::min(slab_, own_got);   // blocks start before the end of the nominal range ..
      avail = own_got;                       // .. and may end in the overlap
      if (start >= got && own_got < slab_ + kOverlap && base + own_got < pread_size_) { c.err = "short read"; file_eof_ = true; return c; }
    } else if (map_) {
      src = map_ + map_pos_;
      got = std::min(slab_, map_size_ - map_pos_);
      avail = map_size_ - map_pos_;          // a block may end past the slab: the mapping has it
    } else {
      comp.alloc(carry_.size() + slab_);
      if (!carry_.empty()) memcpy(comp.data(), carry_.data(), carry_.size());
      got = fread(comp.data() + carry_.size(), 1, slab_, f_);
      avail = carry_.size() + got;
      src = comp.data();
    }
    const size_t scan_end = (map_ || pread_size_) ? got : avail;   // where to stop starting new blocks
    size_t pos = start;
    while (pos + 18 <= avail && pos < scan_end) {
      const uint8_t* h = src + pos;
      if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { c.err = "bad BGZF block"; file_eof_ = true; return c; }
      uint16_t xlen;
      memcpy(&xlen, h + 10, 2);
      if (pos + 12 + xlen > avail) break;          // header extends past what was read: next chunk
      // find the BC subfield (normally the only one, right at h[12..17])
      int bsize = -1;
      for (size_t o = 0; o + 4 <= xlen;) {
        const uint8_t* x = h + 12 + o;
        uint16_t slen;
        memcpy(&slen, x + 2, 2);
        if (x[0] == 'B' && x[1] == 'C' && slen == 2 && o + 6 <= xlen) { uint16_t v; memcpy(&v, x + 4, 2); bsize = v; break; }
        o += 4u + slen;
      }
      if (bsize < 0 || (size_t)bsize + 1 < 12u + xlen + 8u) { c.err = "BGZF block without BC field"; file_eof_ = true; return c; }
      if (pos + (size_t)bsize + 1 > avail) break;  // partial block: next chunk
      const size_t cdata = (size_t)bsize + 1 - 12 - xlen - 8;
      BlockRef b;
      b.coff = pos + 12 + xlen; b.clen = cdata; b.uoff = total;
      memcpy(&b.crc, src + b.coff + cdata, 4);
      memcpy(&b.isize, src + b.coff + cdata + 4, 4);
      if (b.isize > 65536u) { c.err = "bad BGZF block"; file_eof_ = true; return c; }   // (BGZF caps the inflated size)
      total += b.isize;
      blocks.push_back(b);
      pos += (size_t)bsize + 1;
    }
    if (pread_size_) {
      next_off_ = base + pos;
      if (next_off_ >= pread_size_) { c.eof = true; file_eof_ = true; }
      // (stopped early: the file ends inside a block -- the complete blocks before it are still delivered, the error
      // comes when the parser asks for more)
      else if (pos < scan_end) { c.eof = true; file_eof_ = true; c.late_err = "truncated BGZF block"; }
    } else if (map_) {
      map_pos_ += pos;
      if (pos == 0) {                       // nothing complete left: end of file (or a truncated last block)
        c.eof = true;
        file_eof_ = true;
        if (map_pos_ < map_size_) { c.err = "truncated BGZF block"; return c; }
      }
    } else {
      carry_.assign(src + pos, src + avail);
      if (got == 0) {
        c.eof = true;
        file_eof_ = true;
        if (!carry_.empty()) { c.err = "truncated BGZF block"; return c; }
      }
    }
    release();
    const auto ts1 = std::chrono::steady_clock::now();
    // which side inflates this chunk: with percent = 100 the host pool still takes a chunk whenever it would otherwise
    // idle (fewer than two chunks in its queue), everything else goes to the GPU; a smaller percent fixes the share
    bool on_gpu = gpu_.inflate && pread_size_ && !blocks.empty();
    if (on_gpu) {
      if (gpu_percent_ > 100) {
      } else if (gpu_percent_ == 100) {
        if (cpu_inflight_.fetch_add(1) < 2) on_gpu = false; else cpu_inflight_.fetch_sub(1);
      } else {
        on_gpu = (ticket + 1) * (uint64_t)gpu_percent_ / 100 != ticket * (uint64_t)gpu_percent_ / 100;
        if (!on_gpu) cpu_inflight_.fetch_add(1);
      }
    } else cpu_inflight_.fetch_add(1);
    struct CpuDone { std::atomic<int>* c; bool armed; ~CpuDone() { if (armed) c->fetch_sub(1); } } cpu_done{&cpu_inflight_, !on_gpu};
    take_buffer(c.data, total, on_gpu && pin_buffers_);
    const auto ts2 = std::chrono::steady_clock::now();
    if (on_gpu) {
      struct Blk { int64_t coff; int32_t clen; int32_t isize; int64_t uoff; };
      std::vector<Blk> tb(blocks.size());
      for (size_t i = 0; i < blocks.size(); ++i) tb[i] = Blk{(int64_t)blocks[i].coff, (int32_t)blocks[i].clen, (int32_t)blocks[i].isize, (int64_t)blocks[i].uoff};
      GpuObj g = take_gpu_obj(gpu_device_ + (int)(ticket % (uint64_t)gpu_n_devices_));
      int rc = 0;
      if (g.d_cap < total + 256) {
        if (g.d_out) gpu_.device_free(g.dev, g.d_out);
        g.d_out = nullptr; g.d_cap = 0;
        const size_t want = total + total / 4 + 4096;
        rc = gpu_.device_alloc(g.dev, (int64_t)want, &g.d_out);
        if (rc == 0) g.d_cap = want;
      }
      int64_t bad = -1;
      if (rc == 0) rc = gpu_.inflate(&g.h, g.dev, src, (int64_t)avail, tb.data(), (int64_t)tb.size(), g.d_out, c.data.data(), (int64_t)total, &bad);
      put_gpu_obj(g);
      const auto tg = std::chrono::steady_clock::now();
      bool host_instead = false;
      if (rc != 0 && bad >= 0) c.err = "BGZF inflate failed";
      else if (rc != 0) host_instead = true;   // (no memory, no device ...: this chunk goes to the host workers after all)
      else {
        // the footers' CRC32, here on the host (libdeflate's runs at tens of GB/s per core)
        for (const BlockRef& b : blocks) {
          if (b.isize == 0) continue;
          const uint32_t got = BgzfInflater::lib().crc ? BgzfInflater::lib().crc(0, c.data.data() + b.uoff, b.isize)
                                                       : (uint32_t)crc32(0L, c.data.data() + b.uoff, b.isize);
          if (got != b.crc) { c.err = "BGZF block CRC mismatch"; break; }
        }
      }
      const auto tc = std::chrono::steady_clock::now();
      t_gpu_ += (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(tg - ts2).count();
      t_crc_ += (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(tc - tg).count();
      if (!host_instead) { ++n_gpu_chunks_; return c; }
      if (!gpu_warned_.exchange(true)) fprintf(stderr, "[bam_reader] GPU inflate call failed (code %d): the host inflates this chunk\n", rc);
    }
    // groups of 8 blocks per task
    const size_t per = 8, n_tasks = (blocks.size() + per - 1) / per;
    std::vector<std::string> errs(n_tasks);
    const std::function<void(size_t, BgzfInflater&)> work = [&](size_t t, BgzfInflater& inf) {
      for (size_t i = t * per; i < std::min(blocks.size(), (t + 1) * per); ++i) {
        const BlockRef& b = blocks[i];
        if (b.isize == 0) continue;
        if (const char* e = inf.run(src + b.coff, b.clen, c.data.data() + b.uoff, b.isize, b.crc)) { errs[t] = e; return; }
      }
    };
    pool_->run(n_tasks, work);
    const auto ts3 = std::chrono::steady_clock::now();
    auto ns = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count();
    };
    t_scan_ += ns(ts0, ts1); t_buf_ += ns(ts1, ts2); t_inf_ += ns(ts2, ts3);
    for (const std::string& e : errs) if (!e.empty()) { c.err = e; break; }
    return c;
  }

  static constexpr size_t kOverlap = (size_t)128 << 10;   // a block that starts inside a loader's range ends within this
  size_t pread_size_ = 0;          // file size when the loaders pread their own ranges (0: mapping or stream)
  bool pin_buffers_ = true;        // chunk buffers of the GPU inflate path are page-locked (large files)
  size_t next_off_ = 0;            // file offset of the next block to locate (guarded by file_m_)
  std::shared_ptr<FreeList> comp_free_ = std::make_shared<FreeList>();
  std::shared_ptr<Bytes> take_comp() {
    std::shared_ptr<FreeList> fl = comp_free_;
    std::shared_ptr<Bytes> b(new Bytes(), [fl](Bytes* x) {
      { std::lock_guard<std::mutex> lk(fl->m); if (fl->v.size() < 64) fl->v.push_back(std::move(*x)); }
      delete x;
    });
    std::lock_guard<std::mutex> lk(fl->m);
    if (!fl->v.empty()) { b->swap(fl->v.back()); fl->v.pop_back(); }
    return b;
  }
  void prefault_loop() {
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
    const size_t step = (size_t)16 << 20, lead = (size_t)1 << 30;
    size_t done = 0;
    while (done < map_size_) {
      {
        std::unique_lock<std::mutex> lk(file_m_);
        file_cv_.wait(lk, [&] { return prefault_stop_ || done < map_pos_ + lead; });
        if (prefault_stop_) return;
      }
      const size_t n = std::min(step, map_size_ - done);
      if (madvise((void*)(map_ + done), n, MADV_POPULATE_READ) != 0) return;   // (older kernels: faults as before)
      done += n;
    }
  }
  std::thread prefault_;
  bool prefault_stop_ = false;     // (guarded by file_m_)
  size_t ahead_ = 16;              // chunks located / being inflated ahead of the parser (SVDSS_BAM_AHEAD)
  std::shared_ptr<FreeList> free_ = std::make_shared<FreeList>();
  void take_buffer(Bytes& dst, size_t bytes, bool pinned = false) {
    {
      std::lock_guard<std::mutex> lk(free_->m);
      for (size_t i = 0; i < free_->v.size(); ++i)
        if (free_->v[i].cap >= bytes && (!pinned || free_->v[i].p.get_deleter().pinned || free_->v[i].pin_refused)) {
          dst.swap(free_->v[i]); free_->v.erase(free_->v.begin() + (long)i); break;
        }
    }
    const bool fresh = dst.cap < bytes || !dst.p || (pinned && !dst.p.get_deleter().pinned && !dst.pin_refused);
    const auto t0 = std::chrono::steady_clock::now();
    dst.alloc(bytes, pinned);
    if (fresh) {
      ++n_fresh_;
      t_fresh_ += (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    }
  }
  // per-call state of the GPU inflate (stream, device buffers), one per concurrent loader
  struct GpuObj { void* h = nullptr; void* d_out = nullptr; size_t d_cap = 0; int dev = 0; };
  GpuInflateApi gpu_;
  int gpu_device_ = 0, gpu_percent_ = 100, gpu_n_devices_ = 1;
  std::mutex gpu_m_;
  std::vector<GpuObj> gpu_objs_;
  std::atomic<long long> t_gpu_{0}, t_crc_{0}, n_gpu_chunks_{0};
  std::atomic<int> cpu_inflight_{0};   // chunks with the host pool right now
  std::atomic<bool> gpu_warned_{false};
  GpuObj take_gpu_obj(int dev) {
    std::lock_guard<std::mutex> lk(gpu_m_);
    for (size_t i = gpu_objs_.size(); i-- > 0;)
      if (gpu_objs_[i].dev == dev) {
        GpuObj g = gpu_objs_[i];
        gpu_objs_.erase(gpu_objs_.begin() + (long)i);
        return g;
      }
    GpuObj g;
    g.dev = dev;
    return g;
  }
  void put_gpu_obj(const GpuObj& g) { std::lock_guard<std::mutex> lk(gpu_m_); gpu_objs_.push_back(g); }
  std::atomic<long long> t_scan_{0}, t_buf_{0}, t_inf_{0}, n_fresh_{0}, t_fresh_{0};   // SVDSS_DEBUG: nanoseconds per stage
  double t_wait_ = 0;
  std::unique_ptr<InflatePool> pool_;
  FILE* f_;
  int threads_ = 1;
  std::deque<std::future<Chunk>> pending_;
  std::mutex file_m_;
  std::condition_variable file_cv_;
  uint64_t next_ticket_ = 0, n_launched_ = 0;
  bool file_eof_ = false;          // (guarded by file_m_)
  std::vector<uint8_t> carry_;     // partial block at the end of the previous read (guarded by file_m_)
  const uint8_t* map_ = nullptr;   // the whole file, when it could be mapped
  size_t map_size_ = 0, map_pos_ = 0;
  bool reported_ = false;
  void report_impl() {
    if (n_gpu_chunks_.load())
      fprintf(stderr, "[bam_reader] %llu chunks inflated on the GPU (%.3f s summed wall incl. copies), CRC check %.3f s\n",
              (unsigned long long)n_gpu_chunks_.load(), t_gpu_.load() * 1e-9, t_crc_.load() * 1e-9);
    {
      fprintf(stderr, "[bam_reader] %llu chunks: locate %.3f s (under the file lock), buffers %.3f s (%llu fresh: %.3f s summed), inflate %.3f s summed wall, parser waited %.3f s; %d workers\n",
              (unsigned long long)n_launched_, t_scan_.load() * 1e-9, t_buf_.load() * 1e-9, (unsigned long long)n_fresh_.load(), t_fresh_.load() * 1e-9, t_inf_.load() * 1e-9,
              t_wait_ * 1e-9, threads_);
    }
  }
  bool launched_eof_ = false;
  std::string pending_err_;
  bool eof_seen_ = false;
  std::string err_;
  std::vector<std::string> refs_;
  std::vector<int32_t> ref_lens_;
  std::string text_;
  std::shared_ptr<Bytes> chunk_ = std::make_shared<Bytes>();   // the inflated chunk being parsed
  uint64_t chunk_id_ = 0;
  std::vector<uint8_t> buf_;
  size_t upos_ = 0;
};
