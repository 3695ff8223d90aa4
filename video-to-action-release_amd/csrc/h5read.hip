// Minimal native HDF5 reader for the random-action episode file (host code only; no libhdf5, no h5py).
//
// replaces: h5py.File(...)[f'{task}/{ep}']['agentview_image'][:] / ['action'][:] in the reference's loader
//           (diffuser/libero/lb_online_trainer_v7.py:718-780) for the file its generator writes
//           (environment/libero/lb_data/lb_randsam.py:84-104: h5py defaults = "earliest" format):
//             superblock version 0/1, old-style groups (symbol-table message -> v1 B-tree of SNOD nodes + local heap),
//             version-1 object headers (with continuation blocks), dataspace v1/v2, fixed-point / floating-point datatypes
//             (little endian), data layout v1/v2/v3: contiguous, compact, and chunked WITHOUT filters (v1 chunk B-tree).
// Anything else (superblock >= 2, new-style groups, compression / filter pipelines, big-endian, variable-length types for
// DATASETS) is refused with a message -- never guessed at.  Attributes and other header messages are skipped.
// Format source: the public "HDF5 File Format Specification Version 2.0" (sections II.A, III.A-III.D, IV.A.1-IV.A.2).
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

namespace {

struct H5 {
    int fd = -1;
    const uint8_t* map = nullptr;
    size_t size = 0;
    uint64_t base = 0;
    int so = 8, sl = 8;                // size of offsets / lengths
    uint64_t root_btree = 0, root_heap = 0, root_ohdr = 0;
    std::string err;
};

constexpr uint64_t UNDEF = ~0ull;

struct Cur {                            // bounds-checked cursor over the mapping
    const H5* f;
    uint64_t pos;                       // absolute file position (base already added)
    bool ok = true;
    Cur(const H5* f_, uint64_t p) : f(f_), pos(p) {}
    bool need(uint64_t n) {
        if (!ok || pos > f->size || n > f->size - pos) ok = false;
        return ok;
    }
    uint64_t u(int n) {
        if (!need(n)) return 0;
        uint64_t v = 0;
        for (int i = 0; i < n; ++i) v |= (uint64_t)f->map[pos + i] << (8 * i);
        pos += n;
        return v;
    }
    void skip(uint64_t n) {
        if (need(n)) pos += n;
    }
    bool sig(const char* s4) {
        if (!need(4)) return false;
        const bool m = memcmp(f->map + pos, s4, 4) == 0;
        pos += 4;
        return m;
    }
};

struct Msg {
    int type;
    uint64_t pos, size;                 // absolute position of the message data
};

bool fail(H5* f, const std::string& m) {
    f->err = m;
    return false;
}

// every message of a version-1 object header (continuation blocks followed)
bool read_messages(H5* f, uint64_t ohdr_addr, std::vector<Msg>* out) {
    Cur c(f, f->base + ohdr_addr);
    const int ver = (int)c.u(1);
    if (!c.ok) return fail(f, "object header outside the file");
    if (ver != 1) {
        Cur s(f, f->base + ohdr_addr);
        if (s.sig("OHDR")) return fail(f, "version-2 object header (file written with libver='latest'): not supported");
        return fail(f, "unknown object header version " + std::to_string(ver));
    }
    c.skip(1);
    int nmsg = (int)c.u(2);
    c.skip(4);                           // reference count
    uint64_t hsize = c.u(4);
    c.skip(4);                           // pad to 8
    std::vector<std::pair<uint64_t, uint64_t>> blocks = {{c.pos, hsize}};
    for (size_t b = 0; b < blocks.size() && nmsg > 0; ++b) {
        Cur m(f, blocks[b].first);
        const uint64_t end = blocks[b].first + blocks[b].second;
        while (m.pos + 8 <= end && nmsg > 0) {
            const int type = (int)m.u(2);
            const uint64_t sz = m.u(2);
            m.skip(4);                   // flags + reserved
            if (!m.ok || m.pos + sz > end + 7) return fail(f, "truncated object header message");
            --nmsg;
            if (type == 0x10) {          // continuation: offset, length
                Cur k(f, m.pos);
                const uint64_t off = k.u(f->so), len = k.u(f->sl);
                if (!k.ok) return fail(f, "bad continuation message");
                blocks.push_back({f->base + off, len});
            } else if (type != 0) {
                out->push_back({type, m.pos, sz});
            }
            m.skip((sz + 7) & ~7ull);
            if (!m.ok) break;
        }
    }
    return true;
}

const Msg* find(const std::vector<Msg>& ms, int type) {
    for (const auto& m : ms)
        if (m.type == type) return &m;
    return nullptr;
}

std::string heap_name(H5* f, uint64_t heap_addr, uint64_t off) {
    Cur c(f, f->base + heap_addr);
    if (!c.sig("HEAP")) return std::string();
    c.skip(4);
    const uint64_t dsize = c.u(f->sl);
    c.skip(f->sl);
    const uint64_t daddr = c.u(f->so);
    if (!c.ok || off >= dsize) return std::string();
    const uint64_t p = f->base + daddr + off;
    if (p >= f->size) return std::string();
    const size_t n = strnlen((const char*)f->map + p, (size_t)std::min<uint64_t>(f->size - p, dsize - off));
    return std::string((const char*)f->map + p, n);
}

struct Child {
    std::string name;
    uint64_t ohdr;
};

// A malformed file may link B-tree nodes into cycles: besides the depth limit every walk spends from a node budget (a v1 B-tree
// node is >= 24 bytes, so a well-formed file can never hold more nodes than size / 24).
bool spend_node(H5* f, uint64_t* budget) {
    if (*budget == 0) return fail(f, "B-tree visits more nodes than the file can hold (cycle?)");
    --*budget;
    return true;
}

bool walk_group_btree(H5* f, uint64_t node, uint64_t heap, std::vector<Child>* out, int depth, uint64_t* budget) {
    if (depth > 32) return fail(f, "group B-tree deeper than 32 levels");
    if (!spend_node(f, budget)) return false;
    Cur c(f, f->base + node);
    if (c.sig("SNOD")) {
        c.skip(2);
        const int n = (int)c.u(2);
        for (int i = 0; i < n; ++i) {
            const uint64_t name_off = c.u(f->so), oh = c.u(f->so);
            c.skip(4 + 4 + 16);
            if (!c.ok) return fail(f, "truncated symbol table node");
            out->push_back({heap_name(f, heap, name_off), oh});
        }
        return true;
    }
    Cur t(f, f->base + node);
    if (!t.sig("TREE")) return fail(f, "expected a TREE / SNOD node in a group B-tree");
    const int ntype = (int)t.u(1);
    t.skip(1);                           // level
    const int used = (int)t.u(2);
    t.skip(2 * f->so);                   // siblings
    if (ntype != 0) return fail(f, "group B-tree node has the wrong type");
    for (int i = 0; i < used; ++i) {
        t.skip(f->sl);                   // key i
        const uint64_t child = t.u(f->so);
        if (!t.ok) return fail(f, "truncated B-tree node");
        if (!walk_group_btree(f, child, heap, out, depth + 1, budget)) return false;
    }
    return true;
}

bool group_children(H5* f, uint64_t ohdr, std::vector<Child>* out) {
    uint64_t bt, hp;
    if (ohdr == f->root_ohdr && f->root_btree != UNDEF) {
        bt = f->root_btree;
        hp = f->root_heap;
    } else {
        std::vector<Msg> ms;
        if (!read_messages(f, ohdr, &ms)) return false;
        const Msg* st = find(ms, 0x11);
        if (!st) {
            if (find(ms, 0x02) || find(ms, 0x06)) return fail(f, "new-style group (link messages): not supported");
            return fail(f, "not a group");
        }
        Cur c(f, st->pos);
        bt = c.u(f->so);
        hp = c.u(f->so);
        if (!c.ok) return fail(f, "bad symbol table message");
    }
    uint64_t budget = f->size / 24 + 16;
    return walk_group_btree(f, bt, hp, out, 0, &budget);
}

bool resolve(H5* f, const char* path, uint64_t* ohdr) {
    uint64_t cur = f->root_ohdr;
    std::string p(path ? path : "");
    size_t i = 0;
    while (i < p.size()) {
        while (i < p.size() && p[i] == '/') ++i;
        size_t j = i;
        while (j < p.size() && p[j] != '/') ++j;
        if (j == i) break;
        const std::string part = p.substr(i, j - i);
        std::vector<Child> ch;
        if (!group_children(f, cur, &ch)) return false;
        bool hit = false;
        for (const auto& c : ch)
            if (c.name == part) {
                cur = c.ohdr;
                hit = true;
                break;
            }
        if (!hit) return fail(f, "no such object: " + p);
        i = j;
    }
    *ohdr = cur;
    return true;
}

struct DsInfo {
    int tclass = -1, esize = 0, is_signed = 0, ndim = 0, layout = -1;
    uint64_t dims[8] = {0};
    uint64_t nbytes = 0, addr = UNDEF, csize = 0;     // contiguous address / compact data position
    uint64_t chunk_btree = UNDEF;
    uint32_t cdims[9] = {0};
    int cnd = 0;
};

bool dataset_info(H5* f, uint64_t ohdr, DsInfo* d) {
    std::vector<Msg> ms;
    if (!read_messages(f, ohdr, &ms)) return false;
    const Msg *sp = find(ms, 0x01), *dt = find(ms, 0x03), *lay = find(ms, 0x08);
    if (!sp || !dt || !lay) return fail(f, "not a dataset (dataspace / datatype / layout message missing)");
    if (const Msg* fp = find(ms, 0x0B)) {
        Cur c(f, fp->pos);
        c.skip(1);
        if (c.u(1) > 0) return fail(f, "dataset has a filter pipeline (compression): not supported by the native reader");
    }
    {   // dataspace
        Cur c(f, sp->pos);
        const int ver = (int)c.u(1);
        d->ndim = (int)c.u(1);
        const int flags = (int)c.u(1);
        if (ver == 1) c.skip(5);
        else if (ver == 2) c.skip(1);
        else return fail(f, "unknown dataspace message version");
        if (d->ndim > 8) return fail(f, "more than 8 dimensions");
        (void)flags;
        for (int i = 0; i < d->ndim; ++i) d->dims[i] = c.u(f->sl);
        if (!c.ok) return fail(f, "truncated dataspace message");
    }
    {   // datatype
        Cur c(f, dt->pos);
        const int cv = (int)c.u(1);
        const int bits0 = (int)c.u(1);
        c.skip(2);
        d->esize = (int)c.u(4);
        d->tclass = cv & 15;
        if (!c.ok) return fail(f, "truncated datatype message");
        if (d->tclass != 0 && d->tclass != 1) return fail(f, "datatype class " + std::to_string(d->tclass) + " (only fixed / floating point)");
        if (bits0 & 1) return fail(f, "big-endian data: not supported");
        d->is_signed = d->tclass == 0 ? ((bits0 >> 3) & 1) : 1;
    }
    if (d->esize < 1 || d->esize > 16) return fail(f, "unsupported element size");
    uint64_t n = (uint64_t)d->esize;
    for (int i = 0; i < d->ndim; ++i) {
        if (d->dims[i] != 0 && n > (UINT64_MAX >> 1) / d->dims[i]) return fail(f, "dataset size overflows 63 bits");
        n *= d->dims[i];
    }
    d->nbytes = n;
    {   // layout
        Cur c(f, lay->pos);
        const int ver = (int)c.u(1);
        if (ver == 3) {
            d->layout = (int)c.u(1);
            if (d->layout == 0) {
                d->csize = c.u(2);
                d->addr = c.pos;            // absolute
            } else if (d->layout == 1) {
                d->addr = c.u(f->so);
                c.skip(f->sl);
            } else if (d->layout == 2) {
                d->cnd = (int)c.u(1);
                d->chunk_btree = c.u(f->so);
                if (d->cnd < 1 || d->cnd > 9) return fail(f, "bad chunk rank");
                for (int i = 0; i < d->cnd; ++i) d->cdims[i] = (uint32_t)c.u(4);
            } else return fail(f, "unknown layout class");
        } else if (ver == 1 || ver == 2) {
            const int nd = (int)c.u(1);
            d->layout = (int)c.u(1);
            c.skip(5);
            if (d->layout != 0) {
                const uint64_t a = c.u(f->so);
                if (d->layout == 1) d->addr = a;
                else d->chunk_btree = a;
            }
            if (nd > 9) return fail(f, "bad layout rank");
            for (int i = 0; i < nd; ++i) d->cdims[i] = (uint32_t)c.u(4);
            d->cnd = nd;
            if (d->layout == 2) return fail(f, "chunked dataset with a version-1/2 layout message: not supported");
            if (d->layout == 0) {
                d->csize = c.u(4);
                d->addr = c.pos;
            }
        } else return fail(f, "unknown data layout message version " + std::to_string(ver));
        if (!c.ok) return fail(f, "truncated layout message");
    }
    return true;
}

// copy every chunk under `node` into its place of the row-major destination
bool read_chunks(H5* f, const DsInfo& d, uint64_t node, uint8_t* dst, int depth, uint64_t* budget) {
    if (depth > 32) return fail(f, "chunk B-tree deeper than 32 levels");
    if (!spend_node(f, budget)) return false;
    Cur t(f, f->base + node);
    if (!t.sig("TREE")) return fail(f, "expected a TREE node in a chunk B-tree");
    const int ntype = (int)t.u(1), level = (int)t.u(1), used = (int)t.u(2);
    t.skip(2 * f->so);
    if (ntype != 1) return fail(f, "chunk B-tree node has the wrong type");
    const int rank = d.cnd - 1;
    if (rank != d.ndim) return fail(f, "chunk rank does not match the dataspace");
    if (rank < 1 || rank > 8) return fail(f, "chunked dataset needs 1..8 dimensions");
    if (d.cdims[rank] != (uint32_t)d.esize) return fail(f, "chunk element size does not match the datatype");
    for (int k = 0; k < rank; ++k)
        if (d.cdims[k] == 0 || d.dims[k] == 0) return fail(f, "zero-sized chunk or dataset dimension");
    for (int i = 0; i < used; ++i) {
        const uint64_t csz = t.u(4);
        const uint64_t mask = t.u(4);
        uint64_t off[9];
        for (int k = 0; k <= rank; ++k) off[k] = t.u(8);
        const uint64_t child = t.u(f->so);
        if (!t.ok) return fail(f, "truncated chunk B-tree node");
        if (level > 0) {
            if (!read_chunks(f, d, child, dst, depth + 1, budget)) return false;
            continue;
        }
        if (mask != 0) return fail(f, "filtered chunk: not supported");
        uint64_t want = d.esize;
        for (int k = 0; k < rank; ++k) {
            if (want > f->size / d.cdims[k] + 1) return fail(f, "chunk larger than the file");
            want *= d.cdims[k];
        }
        if (csz != want) return fail(f, "chunk size does not match its dimensions (filtered?)");
        if (f->base + child > f->size || csz > f->size - (f->base + child)) return fail(f, "chunk outside the file");
        const uint8_t* src = f->map + f->base + child;
        // iterate the rows of the chunk (all but the last dimension), clipping at the dataset edge
        uint64_t idx[8] = {0};
        const uint64_t row_elems = d.cdims[rank - 1];
        uint64_t nrows = 1;
        for (int k = 0; k < rank - 1; ++k) nrows *= d.cdims[k];
        for (uint64_t r = 0; r < nrows; ++r) {
            uint64_t rem = r;
            bool inside = true;
            for (int k = rank - 2; k >= 0; --k) {
                idx[k] = rem % d.cdims[k];
                rem /= d.cdims[k];
                if (off[k] + idx[k] >= d.dims[k]) inside = false;
            }
            if (!inside || off[rank - 1] >= d.dims[rank - 1]) continue;
            uint64_t lin = 0;
            for (int k = 0; k < rank - 1; ++k) lin = lin * d.dims[k] + off[k] + idx[k];
            lin = lin * d.dims[rank - 1] + off[rank - 1];
            const uint64_t n = std::min<uint64_t>(row_elems, d.dims[rank - 1] - off[rank - 1]);
            memcpy(dst + lin * d.esize, src + r * row_elems * d.esize, n * d.esize);
        }
    }
    return true;
}

}  // namespace

extern "C" {

// HOST functions.  Returns 0 or -1 (then v2a_h5_last_error() explains).  `*out` must be closed with v2a_h5_close.
int v2a_h5_open(const char* path, void** out) {
    if (!path || !out) return -1;
    H5* f = new H5();
    *out = f;
    f->fd = open(path, O_RDONLY);
    if (f->fd < 0) { f->err = std::string("cannot open ") + path; return -1; }
    struct stat st;
    if (fstat(f->fd, &st) != 0 || st.st_size < 64) { f->err = "file too small for an HDF5 superblock"; return -1; }
    f->size = (size_t)st.st_size;
    void* m = mmap(nullptr, f->size, PROT_READ, MAP_PRIVATE, f->fd, 0);
    if (m == MAP_FAILED) { f->err = "mmap failed"; return -1; }
    f->map = (const uint8_t*)m;
    static const uint8_t SIG[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
    uint64_t sb = UNDEF;
    for (uint64_t o = 0; o + 8 <= f->size; o = o ? o * 2 : 512)      // the superblock sits at 0, 512, 1024, 2048, ...
        if (memcmp(f->map + o, SIG, 8) == 0) { sb = o; break; }
    if (sb == UNDEF) { f->err = "not an HDF5 file (signature not found)"; return -1; }
    Cur c(f, sb + 8);
    const int ver = (int)c.u(1);
    if (ver > 1) { f->err = "superblock version " + std::to_string(ver) + " (written with a newer libver): not supported"; return -1; }
    c.skip(4);
    f->so = (int)c.u(1);
    f->sl = (int)c.u(1);
    c.skip(1 + 2 + 2 + 4);
    if (ver == 1) c.skip(4);
    if ((f->so != 8 && f->so != 4) || (f->sl != 8 && f->sl != 4)) { f->err = "unsupported offset / length size"; return -1; }
    f->base = c.u(f->so);
    c.skip(f->so);                        // free-space info
    c.skip(f->so);                        // end of file
    c.skip(f->so);                        // driver info
    c.skip(f->so);                        // root entry: link name offset
    f->root_ohdr = c.u(f->so);
    const uint64_t cache = c.u(4);
    c.skip(4);
    if (cache == 1) {
        f->root_btree = c.u(f->so);
        f->root_heap = c.u(f->so);
    } else {
        f->root_btree = f->root_heap = UNDEF;
    }
    if (!c.ok) { f->err = "truncated superblock"; return -1; }
    if (f->base == 0 && sb != 0) f->base = sb;     // addresses are relative to the superblock when a user block precedes it
    return 0;
}

void v2a_h5_close(void* h) {
    H5* f = (H5*)h;
    if (!f) return;
    if (f->map) munmap((void*)f->map, f->size);
    if (f->fd >= 0) close(f->fd);
    delete f;
}

const char* v2a_h5_last_error(void* h) { return h ? ((H5*)h)->err.c_str() : "null handle"; }

// 1 if `path` names an object, 0 if not, -1 on a format error
int v2a_h5_exists(void* h, const char* path) {
    H5* f = (H5*)h;
    if (!f || !f->map) return -1;
    uint64_t oh;
    f->err.clear();
    if (resolve(f, path, &oh)) return 1;
    // h5py's `path in f` is False both for a missing member and for a path that runs through a dataset
    if (f->err.rfind("no such object", 0) == 0 || f->err == "not a group") { f->err.clear(); return 0; }
    return -1;
}

// names of the members of group `path`, '\n'-separated, into buf (cap bytes incl. the terminator).  Returns the number of members,
// or -1 (error), or -2 (buf too small: call again with a larger one).
long v2a_h5_list(void* h, const char* path, char* buf, size_t cap) {
    H5* f = (H5*)h;
    if (!f || !f->map) return -1;
    uint64_t oh;
    if (!resolve(f, path, &oh)) return -1;
    std::vector<Child> ch;
    if (!group_children(f, oh, &ch)) return -1;
    std::string s;
    for (const auto& c : ch) { s += c.name; s += '\n'; }
    if (s.size() + 1 > cap) return -2;
    memcpy(buf, s.c_str(), s.size() + 1);
    return (long)ch.size();
}

// type_class: 0 integer, 1 floating point.  dims: up to 8 entries.
int v2a_h5_dataset_info(void* h, const char* path, int* type_class, int* elem_size, int* is_signed, int* ndim, long long* dims,
                        long long* nbytes) {
    H5* f = (H5*)h;
    if (!f || !f->map) return -1;
    uint64_t oh;
    DsInfo d;
    if (!resolve(f, path, &oh) || !dataset_info(f, oh, &d)) return -1;
    if (type_class) *type_class = d.tclass;
    if (elem_size) *elem_size = d.esize;
    if (is_signed) *is_signed = d.is_signed;
    if (ndim) *ndim = d.ndim;
    if (dims) for (int i = 0; i < d.ndim; ++i) dims[i] = (long long)d.dims[i];
    if (nbytes) *nbytes = (long long)d.nbytes;
    return 0;
}

// the whole dataset, row-major, into HOST memory `dst` (dst_bytes must equal the dataset's byte size)
int v2a_h5_read(void* h, const char* path, void* dst, size_t dst_bytes) {
    H5* f = (H5*)h;
    if (!f || !f->map || !dst) return -1;
    uint64_t oh;
    DsInfo d;
    if (!resolve(f, path, &oh) || !dataset_info(f, oh, &d)) return -1;
    if (d.nbytes != dst_bytes) { f->err = "destination size does not match the dataset"; return -1; }
    if (d.nbytes == 0) return 0;
    if (d.layout == 1) {
        if (d.addr == UNDEF) { f->err = "contiguous dataset without storage (never written)"; return -1; }
        if (f->base + d.addr > f->size || d.nbytes > f->size - (f->base + d.addr)) { f->err = "dataset outside the file"; return -1; }
        memcpy(dst, f->map + f->base + d.addr, d.nbytes);
        return 0;
    }
    if (d.layout == 0) {
        if (d.csize < d.nbytes || d.addr + d.nbytes > f->size) { f->err = "compact dataset truncated"; return -1; }
        memcpy(dst, f->map + d.addr, d.nbytes);
        return 0;
    }
    if (d.chunk_btree == UNDEF) { f->err = "chunked dataset without an index"; return -1; }
    memset(dst, 0, dst_bytes);            // chunks never written read as the (zero) fill value
    uint64_t budget = f->size / 24 + 16;
    return read_chunks(f, d, d.chunk_btree, (uint8_t*)dst, 0, &budget) ? 0 : -1;
}

}  // extern "C"
