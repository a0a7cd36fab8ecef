// Hdf5Reader.cpp -- see Hdf5Reader.hpp.  Section numbers refer to the "HDF5 File Format Specification Version 3.0".
#include "Hdf5Reader.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <functional>
#include <list>

namespace crfhost {

namespace {

constexpr uint64_t kUndefined = ~uint64_t(0);

[[noreturn]] void fail(const std::string& what) { throw Hdf5Error(what); }

// bounds-checked little-endian cursor over a byte block
struct Buf {
    const unsigned char* p = nullptr;
    size_t n = 0, at = 0;
    Buf(const unsigned char* data, size_t size) : p(data), n(size) {}
    explicit Buf(const std::vector<unsigned char>& v) : p(v.data()), n(v.size()) {}
    void need(size_t k) const {
        if (k > n - std::min(at, n)) fail("truncated structure");
    }
    uint64_t uN(int bytes) {
        need(size_t(bytes));
        uint64_t v = 0;
        for (int i = 0; i < bytes; i++) v |= uint64_t(p[at + size_t(i)]) << (8 * i);
        at += size_t(bytes);
        if (bytes < 8 && bytes > 0) {  // an all-ones field is the "undefined address" of its width
            const uint64_t ones = (uint64_t(1) << (8 * bytes)) - 1;
            if (v == ones && bytes >= 4) return v;  // callers of narrow fields compare themselves
        }
        return v;
    }
    uint8_t u8() { return uint8_t(uN(1)); }
    uint16_t u16() { return uint16_t(uN(2)); }
    uint32_t u32() { return uint32_t(uN(4)); }
    uint64_t u64() { return uN(8); }
    void skip(size_t k) {
        need(k);
        at += k;
    }
    const unsigned char* here() const { return p + at; }
    size_t left() const { return n - std::min(at, n); }
    bool signature(const char* s) {
        need(4);
        const bool ok = std::memcmp(p + at, s, 4) == 0;
        at += 4;
        return ok;
    }
};

int bytesNeeded(uint64_t v) {  // H5VM_limit_enc_size: bytes to hold values up to v
    int bits = 0;
    while (v) {
        bits++;
        v >>= 1;
    }
    return bits / 8 + 1;
}
int log2floor(uint64_t v) {
    int b = -1;
    while (v) {
        b++;
        v >>= 1;
    }
    return b;
}

struct Message {
    uint16_t type = 0;
    uint8_t flags = 0;
    std::vector<unsigned char> data;
};

}  // namespace

struct Hdf5File::Impl {
    FILE* f = nullptr;
    std::string path;
    uint64_t base = 0, fileSize = 0;
    int O = 8, L = 8;  // size of offsets / lengths
    uint64_t rootHeader = kUndefined;
    // root group given as a symbol-table entry with cached B-tree / heap addresses (superblock 0 / 1)
    uint64_t rootBtree = kUndefined, rootHeap = kUndefined;
    // small cache of decoded chunks (a chunk usually spans several of the slabs the loader asks for one after another)
    struct CachedChunk {
        uint64_t address;
        std::vector<unsigned char> bytes;
    };
    mutable std::list<CachedChunk> chunkCache;
    struct ChunkRecord {
        std::vector<uint64_t> offset;  // element offsets of the chunk's first element
        uint64_t address = kUndefined;
        uint64_t size = 0;       // bytes as stored
        uint32_t filterMask = 0;
    };
    mutable std::map<uint64_t, std::vector<ChunkRecord>> chunkIndexCache;  // by dataset header address

    ~Impl() {
        if (f) std::fclose(f);
    }

    std::vector<unsigned char> read(uint64_t address, size_t n) const {
        if (address == kUndefined) fail("read through an undefined address");
        const uint64_t pos = base + address;
        if (pos > fileSize || n > fileSize - pos) fail("structure extends beyond the end of the file (truncated file?)");
        std::vector<unsigned char> v(n);
        if (fseeko(f, off_t(pos), SEEK_SET) != 0 || (n && std::fread(v.data(), 1, n, f) != n)) fail("read failed");
        return v;
    }
    uint64_t offsetField(Buf& b) const {
        const uint64_t v = b.uN(O);
        const uint64_t ones = O >= 8 ? kUndefined : (uint64_t(1) << (8 * O)) - 1;
        return v == ones ? kUndefined : v;
    }
    uint64_t lengthField(Buf& b) const { return b.uN(L); }

    // ---- superblock (III.A) -------------------------------------------------------------------------------------------
    int readSuperblock() {
        static const unsigned char kSig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
        uint64_t at = 0;
        for (;; at = at ? at * 2 : 512) {  // the superblock may follow a user block: 0, 512, 1024, ...
            if (at + 8 > fileSize) fail("no HDF5 signature found");
            base = 0;
            const auto sig = read(at, 8);
            if (std::memcmp(sig.data(), kSig, 8) == 0) break;
        }
        const auto head = read(at, std::min<uint64_t>(fileSize - at, 128));
        Buf b(head);
        b.skip(8);
        const int version = b.u8();
        if (version == 0 || version == 1) {
            b.skip(4);  // free-space version, root symbol table entry version, reserved, shared header version
            O = b.u8();
            L = b.u8();
            b.skip(1);
            b.skip(4);  // group leaf / internal node K
            b.skip(4);  // file consistency flags
            if (version == 1) b.skip(4);  // indexed storage internal node K, reserved
            if (O != 2 && O != 4 && O != 8) fail("unsupported size of offsets");
            if (L != 2 && L != 4 && L != 8) fail("unsupported size of lengths");
            const uint64_t baseAddress = offsetField(b);
            offsetField(b);  // free-space info
            offsetField(b);  // end of file
            offsetField(b);  // driver information
            // root group symbol table entry (III.C)
            offsetField(b);  // link name offset
            rootHeader = offsetField(b);
            const uint32_t cacheType = b.u32();
            b.skip(4);
            if (cacheType == 1) {
                rootBtree = offsetField(b);
                rootHeap = offsetField(b);
            }
            base = at + (baseAddress == kUndefined ? 0 : baseAddress);
        } else if (version == 2 || version == 3) {
            O = b.u8();
            L = b.u8();
            b.skip(1);
            if (O != 2 && O != 4 && O != 8) fail("unsupported size of offsets");
            const uint64_t baseAddress = offsetField(b);
            offsetField(b);  // superblock extension
            offsetField(b);  // end of file
            rootHeader = offsetField(b);
            base = at + (baseAddress == kUndefined ? 0 : baseAddress);
        } else {
            fail("unknown superblock version " + std::to_string(version));
        }
        if (rootHeader == kUndefined) fail("the file has no root group");
        return version;
    }

    // ---- object headers (IV.A.1) --------------------------------------------------------------------------------------
    std::vector<Message> readObjectHeader(uint64_t address) const {
        std::vector<Message> out;
        const auto head = read(address, 16);
        if (std::memcmp(head.data(), "OHDR", 4) == 0) {
            const auto pre = read(address, std::min<uint64_t>(fileSize - base - address, 64));
            Buf b(pre);
            b.skip(4);
            if (b.u8() != 2) fail("unknown object header version");
            const uint8_t flags = b.u8();
            if (flags & 0x20) b.skip(16);
            if (flags & 0x10) b.skip(4);
            const uint64_t chunk0 = b.uN(1 << (flags & 3));
            const bool tracked = (flags & 0x04) != 0;
            std::vector<std::pair<uint64_t, uint64_t>> blocks{{address + b.at, chunk0}};  // message regions
            for (size_t bi = 0; bi < blocks.size(); bi++) {
                const auto block = read(blocks[bi].first, size_t(blocks[bi].second));
                Buf m(block);
                while (m.left() >= size_t(tracked ? 6 : 4)) {
                    Message msg;
                    msg.type = m.u8();
                    const uint16_t size = m.u16();
                    msg.flags = m.u8();
                    if (tracked) m.skip(2);
                    m.need(size);
                    msg.data.assign(m.here(), m.here() + size);
                    m.skip(size);
                    if (msg.type == 0x10) {  // continuation: an OCHK block = signature, messages, checksum
                        Buf c(msg.data);
                        const uint64_t off = offsetField(c), len = lengthField(c);
                        if (len < 8) fail("malformed object header continuation");
                        const auto sig = read(off, 4);
                        if (std::memcmp(sig.data(), "OCHK", 4) != 0) fail("object header continuation without OCHK signature");
                        if (blocks.size() > 4096) fail("object header with an implausible number of continuation blocks");
                        blocks.emplace_back(off + 4, len - 8);
                    } else if (msg.type != 0) {
                        out.push_back(std::move(msg));
                    }
                }
            }
            return out;
        }
        Buf b(head);
        if (b.u8() != 1) fail("unknown object header version");
        b.skip(1);
        const uint16_t count = b.u16();
        b.skip(4);
        const uint32_t size0 = b.u32();
        std::vector<std::pair<uint64_t, uint64_t>> blocks{{address + 16, size0}};
        unsigned seen = 0;
        for (size_t bi = 0; bi < blocks.size() && seen < count; bi++) {
            const auto block = read(blocks[bi].first, size_t(blocks[bi].second));
            Buf m(block);
            while (m.left() >= 8 && seen < count) {
                Message msg;
                msg.type = m.u16();
                const uint16_t size = m.u16();
                msg.flags = m.u8();
                m.skip(3);
                m.need(size);
                msg.data.assign(m.here(), m.here() + size);
                m.skip(size);
                seen++;
                if (msg.type == 0x10) {
                    Buf c(msg.data);
                    const uint64_t off = offsetField(c), len = lengthField(c);
                    if (blocks.size() > 4096) fail("object header with an implausible number of continuation blocks");
                    blocks.emplace_back(off, len);
                } else if (msg.type != 0) {
                    out.push_back(std::move(msg));
                }
            }
        }
        return out;
    }

    // ---- heaps --------------------------------------------------------------------------------------------------------
    // local heap (III.D): returns the data segment
    std::vector<unsigned char> readLocalHeap(uint64_t address) const {
        const auto head = read(address, size_t(8 + 2 * L + O));
        Buf b(head);
        if (!b.signature("HEAP")) fail("local heap signature missing");
        b.skip(4);
        const uint64_t size = lengthField(b);
        lengthField(b);
        const uint64_t data = offsetField(b);
        return read(data, size_t(size));
    }
    // global heap object (III.E)
    std::vector<unsigned char> readGlobalHeapObject(uint64_t collection, uint32_t index) const {
        const auto head = read(collection, size_t(8 + L));
        Buf h(head);
        if (!h.signature("GCOL")) fail("global heap signature missing");
        h.skip(4);
        const uint64_t size = lengthField(h);
        const auto all = read(collection, size_t(size));
        Buf b(all);
        b.skip(size_t(8 + L));
        while (b.left() >= size_t(8 + L)) {
            const uint16_t idx = b.u16();
            b.skip(6);
            const uint64_t len = lengthField(b);
            if (idx == 0) break;  // free space: the rest of the collection
            b.need(size_t(len));
            if (idx == index) return std::vector<unsigned char>(b.here(), b.here() + len);
            b.skip(size_t((len + 7) & ~uint64_t(7)));
        }
        fail("global heap object not found");
    }

    // fractal heap (III.G): managed objects only
    struct FractalHeap {
        uint64_t address = kUndefined;
        int idLength = 0;
        uint8_t flags = 0;
        uint32_t maxManagedObject = 0;
        int width = 0;
        uint64_t startBlock = 0, maxDirectBlock = 0;
        int maxHeapBits = 0;
        uint64_t root = kUndefined;
        int rootRows = 0;
        int offBytes = 0, lenBytes = 0;
    };
    FractalHeap readFractalHeap(uint64_t address) const {
        const auto head = read(address, size_t(22 + 12 * L + 3 * O));
        Buf b(head);
        if (!b.signature("FRHP")) fail("fractal heap signature missing");
        FractalHeap h;
        h.address = address;
        if (b.u8() != 0) fail("unknown fractal heap version");
        h.idLength = b.u16();
        const uint16_t filterLength = b.u16();
        if (filterLength != 0) fail("filtered fractal heaps are not supported");
        h.flags = b.u8();
        h.maxManagedObject = b.u32();
        lengthField(b);   // next huge object id
        offsetField(b);   // v2 B-tree of huge objects
        lengthField(b);   // free space in managed blocks
        offsetField(b);   // free space manager
        for (int i = 0; i < 8; i++) lengthField(b);  // managed space, allocated, iterator offset, #managed, huge size/#, tiny size/#
        h.width = b.u16();
        h.startBlock = lengthField(b);
        h.maxDirectBlock = lengthField(b);
        h.maxHeapBits = b.u16();
        b.skip(2);        // starting rows of the root indirect block
        h.root = offsetField(b);
        h.rootRows = b.u16();
        h.offBytes = (h.maxHeapBits + 7) / 8;
        h.lenBytes = std::min(bytesNeeded(h.maxDirectBlock), bytesNeeded(h.maxManagedObject));
        return h;
    }
    std::vector<unsigned char> readHeapObject(const FractalHeap& h, const unsigned char* id) const {
        const int type = (id[0] >> 4) & 3;
        if (type == 2) {  // tiny: the object is inside the id
            const int len = (id[0] & 0x0F) + 1;
            return std::vector<unsigned char>(id + 1, id + 1 + len);
        }
        if (type != 0) fail("huge fractal heap objects are not supported");
        Buf ib(id + 1, size_t(h.idLength - 1));
        const uint64_t offset = ib.uN(h.offBytes), length = ib.uN(h.lenBytes);
        // locate the direct block that covers `offset` (III.G: doubling table)
        uint64_t blockAddress = h.root, blockOffset = 0, blockSize = h.startBlock;
        if (h.rootRows > 0) {
            const int maxDirectRows = log2floor(h.maxDirectBlock) - log2floor(h.startBlock) + 2;
            const uint64_t rowSpan0 = uint64_t(h.width) * h.startBlock;
            int row = 0;
            uint64_t rowStart = 0;
            blockSize = h.startBlock;
            if (offset >= rowSpan0) {
                row = log2floor(offset / rowSpan0) + 1;
                blockSize = h.startBlock << (row - 1);
                rowStart = rowSpan0 << (row - 1);
            }
            if (row >= h.rootRows || row >= maxDirectRows) fail("fractal heaps with indirect child blocks are not supported");
            const uint64_t col = (offset - rowStart) / blockSize;
            const size_t headSize = size_t(5 + O + h.offBytes);
            const auto ind = read(h.root, headSize + size_t(row * h.width + int(col) + 1) * size_t(O));
            Buf b(ind);
            if (!b.signature("FHIB")) fail("fractal heap indirect block signature missing");
            b.skip(size_t(1 + O + h.offBytes));
            b.skip(size_t(row * h.width + int(col)) * size_t(O));
            blockAddress = offsetField(b);
            blockOffset = rowStart + col * blockSize;
        }
        if (offset < blockOffset || offset + length > blockOffset + blockSize) fail("fractal heap object outside its block");
        return read(blockAddress + (offset - blockOffset), size_t(length));
    }

    // v2 B-tree (III.A.2): every record of the tree, in order
    void walkBtree2(uint64_t address, const std::function<void(int type, const unsigned char* record, int size)>& visit) const {
        const auto head = read(address, size_t(16 + O + 2 + L + 4));
        Buf b(head);
        if (!b.signature("BTHD")) fail("v2 B-tree header signature missing");
        if (b.u8() != 0) fail("unknown v2 B-tree version");
        const int type = b.u8();
        const uint32_t nodeSize = b.u32();
        const int recordSize = b.u16();
        const int depth = b.u16();
        b.skip(2);
        const uint64_t root = offsetField(b);
        const int rootRecords = b.u16();
        if (root == kUndefined || rootRecords == 0) return;
        if (recordSize <= 0 || nodeSize < 16 + uint32_t(recordSize) || nodeSize > (1u << 24) || depth > 16)
            fail("implausible v2 B-tree header");
        uint64_t visited = 0;
        // per level: the widths of a child pointer's "number of records" / "total records" fields
        std::vector<uint64_t> maxRec(size_t(depth) + 1), cumMax(size_t(depth) + 1);
        std::vector<int> nrecBytes(size_t(depth) + 1), cumBytes(size_t(depth) + 1);
        maxRec[0] = (nodeSize - 10) / uint64_t(recordSize);
        cumMax[0] = maxRec[0];
        nrecBytes[0] = bytesNeeded(maxRec[0]);
        cumBytes[0] = 0;
        for (int u = 1; u <= depth; u++) {
            const uint64_t ptr = uint64_t(O) + uint64_t(nrecBytes[size_t(u - 1)]) + uint64_t(cumBytes[size_t(u - 1)]);
            maxRec[size_t(u)] = (nodeSize - (10 + ptr)) / (uint64_t(recordSize) + ptr);
            cumMax[size_t(u)] = (maxRec[size_t(u)] + 1) * cumMax[size_t(u - 1)] + maxRec[size_t(u)];
            nrecBytes[size_t(u)] = bytesNeeded(maxRec[size_t(u)]);
            cumBytes[size_t(u)] = bytesNeeded(cumMax[size_t(u)]);
        }
        std::function<void(uint64_t, int, int)> node = [&](uint64_t at, int records, int level) {
            if (++visited > (1u << 20) || uint64_t(records) > maxRec[size_t(level)]) fail("damaged v2 B-tree");
            const auto raw = read(at, nodeSize);
            Buf n(raw);
            if (!n.signature(level == 0 ? "BTLF" : "BTIN")) fail("v2 B-tree node signature missing");
            n.skip(2);
            const unsigned char* recs = n.here();
            n.skip(size_t(records) * size_t(recordSize));
            if (level == 0) {
                for (int i = 0; i < records; i++) visit(type, recs + size_t(i) * size_t(recordSize), recordSize);
                return;
            }
            std::vector<std::pair<uint64_t, int>> children;
            for (int i = 0; i <= records; i++) {
                const uint64_t child = offsetField(n);
                const int childRecords = int(n.uN(nrecBytes[size_t(level - 1)]));
                if (level > 1) n.skip(size_t(cumBytes[size_t(level - 1)]));
                children.emplace_back(child, childRecords);
            }
            for (int i = 0; i <= records; i++) {
                node(children[size_t(i)].first, children[size_t(i)].second, level - 1);
                if (i < records) visit(type, recs + size_t(i) * size_t(recordSize), recordSize);
            }
        };
        node(root, rootRecords, depth);
    }

    // ---- messages -----------------------------------------------------------------------------------------------------
    // datatype (IV.A.2.d); returns the bytes consumed
    size_t parseDatatype(const unsigned char* p, size_t n, Hdf5Datatype* t, int depth = 0) const {
        if (depth > 8) fail("datatype nested too deeply");
        Buf b(p, n);
        const uint8_t cv = b.u8();
        const int cls = cv & 0x0F;
        const uint8_t bits0 = b.u8(), bits1 = b.u8();
        b.skip(1);
        t->size = b.u32();
        t->cls = Hdf5Datatype::OTHER;
        switch (cls) {
            case 0:
                t->cls = Hdf5Datatype::FIXED;
                t->bigEndian = bits0 & 1;
                t->isSigned = (bits0 & 8) != 0;
                b.skip(4);
                break;
            case 1:
                t->cls = Hdf5Datatype::FLOAT;
                t->bigEndian = bits0 & 1;
                if (bits0 & 0x40) fail("VAX floating point is not supported");
                b.skip(12);
                break;
            case 3: t->cls = Hdf5Datatype::STRING; break;
            case 7: t->cls = Hdf5Datatype::REFERENCE; break;
            case 9: {
                t->cls = Hdf5Datatype::VLEN;
                t->vlenString = (bits0 & 0x0F) == 1;
                (void)bits1;
                t->base = std::make_shared<Hdf5Datatype>();
                b.skip(parseDatatype(b.here(), b.left(), t->base.get(), depth + 1));
                break;
            }
            case 6: t->cls = Hdf5Datatype::COMPOUND; break;  // REFERENCE_LIST: size known from the message, members not needed
            default: break;
        }
        return b.at;
    }
    // dataspace (IV.A.2.b)
    std::vector<uint64_t> parseDataspace(const unsigned char* p, size_t n) const {
        Buf b(p, n);
        const int version = b.u8();
        const int rank = b.u8();
        const uint8_t flags = b.u8();
        if (version == 1) {
            b.skip(5);
        } else if (version == 2) {
            const int type = b.u8();
            if (type == 2) return {0};  // null dataspace: no elements
        } else {
            fail("unknown dataspace version");
        }
        (void)flags;
        std::vector<uint64_t> shape(static_cast<size_t>(rank));
        for (int i = 0; i < rank; i++) shape[size_t(i)] = lengthField(b);
        return shape;
    }

    double numberAt(const Hdf5Datatype& t, const unsigned char* p) const {
        unsigned char v[8] = {0};
        for (uint32_t i = 0; i < t.size && i < 8; i++) v[i] = t.bigEndian ? p[t.size - 1 - i] : p[i];
        if (t.cls == Hdf5Datatype::FLOAT) {
            if (t.size == 4) {
                float f;
                std::memcpy(&f, v, 4);
                return f;
            }
            if (t.size == 8) {
                double d;
                std::memcpy(&d, v, 8);
                return d;
            }
            fail("floating-point attribute of unsupported size");
        }
        uint64_t u = 0;
        std::memcpy(&u, v, 8);
        if (t.isSigned && t.size < 8 && (u >> (8 * t.size - 1)) & 1) u |= ~uint64_t(0) << (8 * t.size);
        return t.isSigned ? double(int64_t(u)) : double(u);
    }

    // attribute message (IV.A.2.m)
    Hdf5Attribute parseAttribute(const std::vector<unsigned char>& data) const {
        Buf b(data);
        const int version = b.u8();
        const uint8_t flags = b.u8();
        const uint16_t nameSize = b.u16(), typeSize = b.u16(), spaceSize = b.u16();
        if (version < 1 || version > 3) fail("unknown attribute message version");
        if (version >= 2 && (flags & 3)) fail("attributes with shared datatypes / dataspaces are not supported");
        if (version == 3) b.skip(1);
        auto padded = [&](size_t k) { return version == 1 ? (k + 7) & ~size_t(7) : k; };
        Hdf5Attribute a;
        b.need(nameSize);
        a.name.assign(reinterpret_cast<const char*>(b.here()), strnlen(reinterpret_cast<const char*>(b.here()), nameSize));
        b.skip(padded(nameSize));
        b.need(typeSize);
        parseDatatype(b.here(), typeSize, &a.type);
        b.skip(padded(typeSize));
        b.need(spaceSize);
        a.shape = parseDataspace(b.here(), spaceSize);
        b.skip(padded(spaceSize));
        uint64_t count = 1;
        for (uint64_t d : a.shape) count *= d;
        const uint64_t bytes = std::min<uint64_t>(count * a.type.size, b.left());
        a.raw.assign(b.here(), b.here() + bytes);
        const uint64_t have = a.type.size ? bytes / a.type.size : 0;
        if (a.type.cls == Hdf5Datatype::STRING && have >= 1) {
            a.isString = true;
            a.text.assign(reinterpret_cast<const char*>(a.raw.data()), strnlen(reinterpret_cast<const char*>(a.raw.data()), a.type.size));
            while (!a.text.empty() && a.text.back() == ' ') a.text.pop_back();  // space-padded strings
        } else if (a.type.cls == Hdf5Datatype::FIXED || a.type.cls == Hdf5Datatype::FLOAT) {
            a.isNumeric = true;
            for (uint64_t i = 0; i < have; i++) a.numbers.push_back(numberAt(a.type, a.raw.data() + i * a.type.size));
        } else if (a.type.cls == Hdf5Datatype::VLEN) {
            // element: length (4), global heap collection address (O), object index (4)   (IV.A.2.d, class 9)
            for (uint64_t i = 0; i < have; i++) {
                Buf e(a.raw.data() + i * a.type.size, a.type.size);
                const uint32_t len = e.u32();
                const uint64_t collection = offsetField(e);
                const uint32_t index = e.u32();
                if (len == 0 || collection == kUndefined || collection == 0) {
                    if (!a.type.vlenString) a.references.push_back(kUndefined);
                    continue;
                }
                const auto obj = readGlobalHeapObject(collection, index);
                if (a.type.vlenString) {
                    if (i == 0) {
                        a.isString = true;
                        a.text.assign(reinterpret_cast<const char*>(obj.data()), std::min<size_t>(len, obj.size()));
                    }
                } else if (a.type.base && a.type.base->cls == Hdf5Datatype::REFERENCE && obj.size() >= size_t(O)) {
                    Buf r(obj);
                    a.references.push_back(offsetField(r));
                }
            }
        }
        return a;
    }

    // link message (IV.A.2.g): name + object header address of a hard link (others: address undefined); *order = the
    // link's creation order when the group tracks it
    std::pair<std::string, uint64_t> parseLink(const std::vector<unsigned char>& data, int64_t* order) const {
        Buf b(data);
        if (b.u8() != 1) fail("unknown link message version");
        const uint8_t flags = b.u8();
        int type = 0;
        if (flags & 0x08) type = b.u8();
        *order = -1;
        if (flags & 0x04) *order = int64_t(b.u64());
        if (flags & 0x10) b.skip(1);
        const uint64_t nameLength = b.uN(1 << (flags & 3));
        b.need(size_t(nameLength));
        std::string name(reinterpret_cast<const char*>(b.here()), size_t(nameLength));
        b.skip(size_t(nameLength));
        uint64_t address = kUndefined;
        if (type == 0) address = offsetField(b);
        return {name, address};
    }

    // every (name, object header address) of a group, whichever way it is stored
    std::vector<std::pair<std::string, uint64_t>> readGroupLinks(const std::vector<Message>& messages, uint64_t btree,
                                                                 uint64_t heap) const {
        std::vector<std::pair<std::string, uint64_t>> links;
        std::vector<int64_t> orders;  // creation order per link, -1 when not tracked
        auto addLink = [&](const std::vector<unsigned char>& body) {
            int64_t order = -1;
            links.push_back(parseLink(body, &order));
            orders.push_back(order);
        };
        for (const Message& m : messages) {
            if (m.type == 0x11) {  // symbol table message
                Buf b(m.data);
                btree = offsetField(b);
                heap = offsetField(b);
            } else if (m.type == 0x06) {
                addLink(m.data);
            } else if (m.type == 0x02) {  // link info: dense storage when the heap address is defined
                Buf b(m.data);
                if (b.u8() != 0) fail("unknown link info version");
                const uint8_t flags = b.u8();
                if (flags & 1) b.skip(8);
                const uint64_t heapAddress = offsetField(b);
                const uint64_t nameIndex = offsetField(b);
                if (heapAddress != kUndefined && nameIndex != kUndefined) {
                    const FractalHeap fh = readFractalHeap(heapAddress);
                    walkBtree2(nameIndex, [&](int type, const unsigned char* rec, int size) {
                        if (type != 5 || size < 4 + fh.idLength) fail("unexpected record type in a group's name index");
                        addLink(readHeapObject(fh, rec + 4));
                    });
                }
            }
        }
        if (btree != kUndefined && heap != kUndefined) {  // old-style group (III.B / III.C)
            const auto names = readLocalHeap(heap);
            uint64_t visited = 0;
            std::function<void(uint64_t)> node = [&](uint64_t at) {
                if (++visited > (1u << 20)) fail("damaged group B-tree (cycle?)");
                const auto head = read(at, size_t(8 + 2 * O));
                Buf h(head);
                if (h.signature("TREE")) {
                    if (h.u8() != 0) fail("group B-tree of the wrong node type");
                    h.skip(1);
                    const int entries = h.u16();
                    const auto body = read(at + 8 + 2 * uint64_t(O), size_t(entries) * size_t(L + O) + size_t(L));
                    Buf b(body);
                    for (int i = 0; i < entries; i++) {
                        lengthField(b);
                        node(offsetField(b));
                    }
                    return;
                }
                Buf s(head);
                if (!s.signature("SNOD")) fail("group node signature missing");
                s.skip(2);
                const int symbols = s.u16();
                const auto body = read(at + 8, size_t(symbols) * size_t(2 * O + 24));
                Buf b(body);
                for (int i = 0; i < symbols; i++) {
                    const uint64_t nameOffset = offsetField(b);
                    const uint64_t header = offsetField(b);
                    b.skip(24);
                    if (nameOffset >= names.size()) fail("symbol name outside the local heap");
                    const char* s0 = reinterpret_cast<const char*>(names.data() + nameOffset);
                    links.emplace_back(std::string(s0, strnlen(s0, names.size() - size_t(nameOffset))), header);
                }
            };
            node(btree);
        }
        // netcdf-c numbers variables and dimensions in creation order (it requires the tracking it sets up with
        // H5Pset_link_creation_order): present the links that way whenever every link carries its order
        if (!links.empty() && orders.size() == links.size() && std::all_of(orders.begin(), orders.end(), [](int64_t o) { return o >= 0; })) {
            std::vector<size_t> perm(links.size());
            for (size_t i = 0; i < perm.size(); i++) perm[i] = i;
            std::stable_sort(perm.begin(), perm.end(), [&](size_t a, size_t b) { return orders[a] < orders[b]; });
            std::vector<std::pair<std::string, uint64_t>> sorted;
            for (size_t i : perm) sorted.push_back(links[i]);
            links.swap(sorted);
        }
        return links;
    }

    void collectAttributes(const std::vector<Message>& messages, std::map<std::string, Hdf5Attribute>* out) const {
        for (const Message& m : messages) {
            if (m.type == 0x0C) {
                if (m.flags & 0x02) fail("shared attribute messages are not supported");
                Hdf5Attribute a = parseAttribute(m.data);
                (*out)[a.name] = std::move(a);
            } else if (m.type == 0x15) {  // attribute info: dense attribute storage
                Buf b(m.data);
                if (b.u8() != 0) fail("unknown attribute info version");
                const uint8_t flags = b.u8();
                if (flags & 1) b.skip(2);
                const uint64_t heapAddress = offsetField(b);
                const uint64_t nameIndex = offsetField(b);
                if (heapAddress == kUndefined || nameIndex == kUndefined) continue;
                const FractalHeap fh = readFractalHeap(heapAddress);
                walkBtree2(nameIndex, [&](int type, const unsigned char* rec, int size) {
                    if (type != 8 || size < fh.idLength) fail("unexpected record type in an attribute name index");
                    Hdf5Attribute a = parseAttribute(readHeapObject(fh, rec));
                    (*out)[a.name] = std::move(a);
                });
            }
        }
    }

    // ---- datasets -----------------------------------------------------------------------------------------------------
    bool parseDataset(const std::string& name, uint64_t address, const std::vector<Message>& messages, Hdf5Dataset* ds) const {
        bool haveSpace = false, haveType = false, haveLayout = false;
        ds->name = name;
        ds->headerAddress = address;
        for (const Message& m : messages) {
            if ((m.flags & 0x02) && (m.type == 0x01 || m.type == 0x03 || m.type == 0x0B))
                fail("dataset \"" + name + "\": shared (committed) datatype / dataspace / filter messages are not supported");
            if (m.type == 0x01) {
                ds->shape = parseDataspace(m.data.data(), m.data.size());
                haveSpace = true;
            } else if (m.type == 0x03) {
                parseDatatype(m.data.data(), m.data.size(), &ds->type);
                haveType = true;
            } else if (m.type == 0x08) {
                Buf b(m.data);
                const int version = b.u8();
                if (version != 3 && version != 4) fail("dataset \"" + name + "\": data layout message version " + std::to_string(version) + " is not supported");
                ds->layoutClass = b.u8();
                if (ds->layoutClass == 0) {
                    const uint16_t size = b.u16();
                    b.need(size);
                    ds->compactData.assign(b.here(), b.here() + size);
                } else if (ds->layoutClass == 1) {
                    ds->dataAddress = offsetField(b);
                    ds->dataSize = lengthField(b);
                } else if (ds->layoutClass == 2 && version == 3) {
                    const int dims = b.u8();
                    ds->chunkIndexAddress = offsetField(b);
                    for (int i = 0; i < dims; i++) ds->chunkShape.push_back(b.u32());
                    if (!ds->chunkShape.empty()) ds->chunkShape.pop_back();  // the last "dimension" is the element size
                    ds->chunkIndexType = 0;
                } else if (ds->layoutClass == 2) {
                    const uint8_t flags = b.u8();
                    const int dims = b.u8();
                    const int enc = b.u8();
                    for (int i = 0; i < dims; i++) ds->chunkShape.push_back(b.uN(enc));
                    if (!ds->chunkShape.empty()) ds->chunkShape.pop_back();
                    ds->chunkIndexType = b.u8();
                    if (ds->chunkIndexType == 1) {
                        if (flags & 0x02) {
                            ds->singleChunkSize = lengthField(b);
                            ds->singleChunkMask = b.u32();
                        }
                    } else if (ds->chunkIndexType == 3) {
                        b.skip(1);  // page bits (repeated in the fixed array header)
                    } else if (ds->chunkIndexType == 4) {
                        b.skip(5);
                    } else if (ds->chunkIndexType == 5) {
                        b.skip(6);
                    }
                    ds->chunkIndexAddress = offsetField(b);
                } else {
                    fail("dataset \"" + name + "\": virtual / unknown storage layout");
                }
                haveLayout = true;
            } else if (m.type == 0x0B) {
                Buf b(m.data);
                const int version = b.u8();
                const int count = b.u8();
                if (version == 1) b.skip(6);
                else if (version != 2) fail("unknown filter pipeline version");
                for (int i = 0; i < count; i++) {
                    Hdf5Filter f;
                    f.id = b.u16();
                    uint16_t nameLength = 0;
                    if (version == 1 || f.id >= 256) nameLength = b.u16();
                    b.skip(2);  // flags
                    const uint16_t values = b.u16();
                    b.skip(version == 1 ? (size_t(nameLength) + 7) & ~size_t(7) : nameLength);
                    for (int v = 0; v < values; v++) f.clientData.push_back(b.u32());
                    if (version == 1 && (values & 1)) b.skip(4);
                    ds->filters.push_back(f);
                }
            }
        }
        if (!haveSpace || !haveType || !haveLayout) return false;  // a group or a committed datatype, not a dataset
        // plausibility against the file's size (a damaged dimension must not turn into a huge allocation downstream):
        // contiguous data lies in the file as it is; deflate expands by at most ~1032:1
        uint64_t elements = 1;
        for (uint64_t d : ds->shape) {
            if (d && elements > (uint64_t(1) << 48) / d) fail("dataset \"" + name + "\": implausible shape");
            elements *= d;
        }
        const uint64_t bytes = elements * ds->type.size;
        if (ds->layoutClass == 1 && ds->dataAddress != kUndefined &&
            (ds->dataSize != bytes || ds->dataAddress > fileSize || bytes > fileSize - ds->dataAddress))
            fail("dataset \"" + name + "\": contiguous storage does not match its shape (damaged file?)");
        if (ds->layoutClass == 2 && bytes / 1100 > fileSize) fail("dataset \"" + name + "\": shape larger than the file can hold (damaged file?)");
        collectAttributes(messages, &ds->attributes);
        return true;
    }

    // the chunk records of a dataset (cached)
    const std::vector<ChunkRecord>& chunkIndex(const Hdf5Dataset& ds) const {
        auto it = chunkIndexCache.find(ds.headerAddress);
        if (it != chunkIndexCache.end()) return it->second;
        std::vector<ChunkRecord> records;
        const size_t rank = ds.shape.size();
        uint64_t chunkBytes = ds.type.size;
        if (ds.chunkShape.size() != rank) fail("dataset \"" + ds.name + "\": chunk rank does not match");
        for (uint64_t c : ds.chunkShape) {
            if (c == 0 || c > (uint64_t(1) << 32) || chunkBytes > (uint64_t(1) << 40) / c) fail("dataset \"" + ds.name + "\": implausible chunk shape");
            chunkBytes *= c;
        }
        if (ds.chunkIndexAddress == kUndefined) {
            // no chunk was ever written: every element reads as the fill value (left to the caller: zeros)
        } else if (ds.chunkIndexType == 0) {  // v1 B-tree, node type 1 (III.A.1)
            uint64_t visited = 0;
            std::function<void(uint64_t)> node = [&](uint64_t at) {
                if (++visited > (1u << 22)) fail("damaged chunk B-tree (cycle?)");
                const auto head = read(at, size_t(8 + 2 * O));
                Buf h(head);
                if (!h.signature("TREE")) fail("chunk B-tree signature missing");
                if (h.u8() != 1) fail("chunk B-tree of the wrong node type");
                const int level = h.u8();
                const int entries = h.u16();
                const size_t keySize = 8 + 8 * (rank + 1);
                const auto body = read(at + 8 + 2 * uint64_t(O), size_t(entries) * (keySize + size_t(O)) + keySize);
                Buf b(body);
                for (int i = 0; i < entries; i++) {
                    ChunkRecord r;
                    r.size = b.u32();
                    r.filterMask = b.u32();
                    for (size_t d = 0; d < rank; d++) r.offset.push_back(b.u64());
                    b.skip(8);
                    r.address = offsetField(b);
                    if (level > 0) node(r.address);
                    else records.push_back(std::move(r));
                }
            };
            node(ds.chunkIndexAddress);
        } else if (ds.chunkIndexType == 1) {  // single chunk
            ChunkRecord r;
            r.offset.assign(rank, 0);
            r.address = ds.chunkIndexAddress;
            r.size = ds.singleChunkSize ? ds.singleChunkSize : chunkBytes;
            r.filterMask = ds.singleChunkMask;
            records.push_back(r);
        } else if (ds.chunkIndexType == 2 || ds.chunkIndexType == 3) {
            // implicit: unfiltered chunks back to back in index order; fixed array (VII.C): one element per chunk
            std::vector<uint64_t> counts(rank);
            uint64_t total = 1;
            for (size_t d = 0; d < rank; d++) {
                counts[d] = (ds.shape[d] + ds.chunkShape[d] - 1) / ds.chunkShape[d];
                if (counts[d] > (uint64_t(1) << 26) || total > (uint64_t(1) << 26)) fail("dataset \"" + ds.name + "\": implausible chunk grid");
                total *= counts[d];
            }
            if (total > (uint64_t(1) << 26)) fail("dataset \"" + ds.name + "\": implausible chunk grid");
            std::vector<unsigned char> elements;
            int entrySize = O;
            bool filtered = false;
            if (ds.chunkIndexType == 3) {
                const auto head = read(ds.chunkIndexAddress, size_t(12 + L + O));
                Buf h(head);
                if (!h.signature("FAHD")) fail("fixed array header signature missing");
                h.skip(1);
                filtered = h.u8() == 1;
                entrySize = h.u8();
                const int pageBits = h.u8();
                const uint64_t entries = lengthField(h);
                const uint64_t block = offsetField(h);
                if (entries != total) fail("fixed array size does not match the chunk grid");
                if (entries > (uint64_t(1) << pageBits)) fail("dataset \"" + ds.name + "\": paged fixed-array chunk indexes are not supported");
                if (block == kUndefined) {
                    chunkIndexCache[ds.headerAddress] = records;
                    return chunkIndexCache[ds.headerAddress];
                }
                const auto body = read(block, size_t(6 + O) + size_t(entries) * size_t(entrySize));
                Buf b(body);
                if (!b.signature("FADB")) fail("fixed array data block signature missing");
                b.skip(size_t(2 + O));
                elements.assign(b.here(), b.here() + size_t(entries) * size_t(entrySize));
            }
            std::vector<uint64_t> idx(rank, 0);
            for (uint64_t i = 0; i < total; i++) {
                ChunkRecord r;
                for (size_t d = 0; d < rank; d++) r.offset.push_back(idx[d] * ds.chunkShape[d]);
                if (ds.chunkIndexType == 2) {
                    r.address = ds.chunkIndexAddress + i * chunkBytes;
                    r.size = chunkBytes;
                } else {
                    Buf e(elements.data() + size_t(i) * size_t(entrySize), size_t(entrySize));
                    r.address = offsetField(e);
                    r.size = chunkBytes;
                    if (filtered) {
                        r.size = e.uN(entrySize - O - 4);
                        r.filterMask = e.u32();
                    }
                }
                if (r.address != kUndefined) records.push_back(std::move(r));
                for (size_t d = rank; d-- > 0;) {
                    if (++idx[d] < counts[d]) break;
                    idx[d] = 0;
                }
            }
        } else {
            fail("dataset \"" + ds.name + "\": chunk index type " + std::to_string(ds.chunkIndexType) +
                 " (extensible array / v2 B-tree: HDF5 1.10 'latest' format with an unlimited dimension) is not supported");
        }
        return chunkIndexCache[ds.headerAddress] = std::move(records);
    }

    // one chunk as stored -> its element bytes (filters undone in reverse order, IV.A.2.l)
    const std::vector<unsigned char>& loadChunk(const Hdf5Dataset& ds, const ChunkRecord& r, uint64_t chunkBytes) const {
        for (auto it = chunkCache.begin(); it != chunkCache.end(); ++it)
            if (it->address == r.address) {
                chunkCache.splice(chunkCache.begin(), chunkCache, it);
                return chunkCache.front().bytes;
            }
        if (r.size > chunkBytes + (chunkBytes >> 2) + 4096) fail("dataset \"" + ds.name + "\": a stored chunk is larger than its shape allows");
        std::vector<unsigned char> bytes = read(r.address, size_t(r.size));
        for (size_t fi = ds.filters.size(); fi-- > 0;) {
            if (r.filterMask & (1u << fi)) continue;  // the filter was skipped for this chunk
            const Hdf5Filter& f = ds.filters[fi];
            if (f.id == 1) {  // deflate
                // the plain size is known: the chunk's elements (+ a checksum when fletcher32 precedes deflate)
                std::vector<unsigned char> plain(size_t(chunkBytes) + 64);
                uLongf n = uLongf(plain.size());
                if (uncompress(plain.data(), &n, bytes.data(), uLong(bytes.size())) != Z_OK)
                    fail("dataset \"" + ds.name + "\": inflating a chunk failed");
                plain.resize(n);
                bytes.swap(plain);
            } else if (f.id == 2) {  // shuffle: byte planes -> elements
                const size_t es = f.clientData.empty() ? ds.type.size : f.clientData[0];
                if (es > 1) {
                    const size_t count = bytes.size() / es;
                    std::vector<unsigned char> plain(bytes.size());
                    for (size_t j = 0; j < es; j++)
                        for (size_t i = 0; i < count; i++) plain[i * es + j] = bytes[j * count + i];
                    for (size_t i = count * es; i < bytes.size(); i++) plain[i] = bytes[i];
                    bytes.swap(plain);
                }
            } else if (f.id == 3) {  // fletcher32: a trailing checksum
                if (bytes.size() < 4) fail("chunk shorter than its checksum");
                bytes.resize(bytes.size() - 4);
            } else {
                fail("dataset \"" + ds.name + "\": filter " + std::to_string(f.id) + " is not supported (deflate, shuffle, fletcher32 are)");
            }
        }
        if (bytes.size() < chunkBytes) fail("dataset \"" + ds.name + "\": a chunk is shorter than its shape");
        chunkCache.push_front({r.address, std::move(bytes)});
        if (chunkCache.size() > 16) chunkCache.pop_back();
        return chunkCache.front().bytes;
    }
};

bool Hdf5Dataset::isDimensionScale() const {
    auto it = attributes.find("CLASS");
    return it != attributes.end() && it->second.isString && it->second.text == "DIMENSION_SCALE";
}

Hdf5File::Hdf5File(const std::string& path) : impl_(new Impl) {
    Impl& h = *impl_;
    h.path = path;
    h.f = std::fopen(path.c_str(), "rb");
    if (!h.f) fail("file could not be opened");
    if (fseeko(h.f, 0, SEEK_END) != 0) fail("seek failed");
    h.fileSize = uint64_t(ftello(h.f));
    superblockVersion_ = h.readSuperblock();
    const auto rootMessages = h.readObjectHeader(h.rootHeader);
    h.collectAttributes(rootMessages, &rootAttributes_);
    for (const auto& link : h.readGroupLinks(rootMessages, h.rootBtree, h.rootHeap)) {
        if (link.second == kUndefined) continue;  // soft / external link
        Hdf5Dataset ds;
        if (h.parseDataset(link.first, link.second, h.readObjectHeader(link.second), &ds)) datasets_.push_back(std::move(ds));
    }
}

Hdf5File::~Hdf5File() = default;

void Hdf5File::readFloats(const Hdf5Dataset& ds, const std::vector<uint64_t>& start, const std::vector<uint64_t>& count,
                          float* out) const {
    const Impl& h = *impl_;
    const size_t rank = ds.shape.size();
    if (start.size() != rank || count.size() != rank) fail("hyperslab rank does not match the dataset");
    if (ds.type.cls != Hdf5Datatype::FLOAT || (ds.type.size != 4 && ds.type.size != 8))
        fail("dataset \"" + ds.name + "\" is not float32 / float64");
    uint64_t total = 1;
    for (size_t d = 0; d < rank; d++) {
        if (start[d] > ds.shape[d] || count[d] > ds.shape[d] - start[d]) fail("hyperslab outside dataset \"" + ds.name + "\"");
        if (count[d] && total > (uint64_t(1) << 40) / count[d]) fail("hyperslab too large");
        total *= count[d];
    }
    if (total == 0) return;
    const uint32_t es = ds.type.size;
    auto convert = [&](const unsigned char* src, float* dst, uint64_t n) {
        for (uint64_t i = 0; i < n; i++) {
            unsigned char v[8];
            for (uint32_t k = 0; k < es; k++) v[k] = ds.type.bigEndian ? src[i * es + (es - 1 - k)] : src[i * es + k];
            if (es == 4) {
                std::memcpy(dst + i, v, 4);
            } else {
                double d;
                std::memcpy(&d, v, 8);
                dst[i] = float(d);
            }
        }
    };
    // copies the intersection of the hyperslab with the box [boxStart, boxStart + boxShape) whose elements lie row-major
    // in `bytes`
    auto copyBox = [&](const unsigned char* bytes, const std::vector<uint64_t>& boxStart, const std::vector<uint64_t>& boxShape) {
        std::vector<uint64_t> lo(rank), hi(rank);
        for (size_t d = 0; d < rank; d++) {
            lo[d] = std::max(start[d], boxStart[d]);
            hi[d] = std::min({start[d] + count[d], boxStart[d] + boxShape[d], ds.shape[d]});
            if (lo[d] >= hi[d]) return;
        }
        std::vector<uint64_t> idx(lo);
        const uint64_t run = hi[rank - 1] - lo[rank - 1];
        for (;;) {
            uint64_t src = 0, dst = 0;
            for (size_t d = 0; d < rank; d++) {
                src = src * boxShape[d] + (idx[d] - boxStart[d]);
                dst = dst * count[d] + (idx[d] - start[d]);
            }
            convert(bytes + src * es, out + dst, run);
            size_t d = rank - 1;
            for (;;) {
                if (d == 0) return;
                d--;
                if (++idx[d] < hi[d]) break;
                idx[d] = lo[d];
                if (d == 0) return;
            }
        }
    };
    const std::vector<uint64_t> zero(rank, 0);
    if (rank == 0) fail("scalar datasets are not volumes");
    if (ds.layoutClass == 0) {
        copyBox(ds.compactData.data(), zero, ds.shape);
    } else if (ds.layoutClass == 1) {
        if (ds.dataAddress == kUndefined) {  // never written: the fill value (HDF5's default is zero)
            std::fill(out, out + total, 0.0f);
            return;
        }
        // contiguous: read the rows the hyperslab covers along the leading dimension
        uint64_t rowElems = 1;
        for (size_t d = 1; d < rank; d++) rowElems *= ds.shape[d];
        const auto bytes = h.read(ds.dataAddress + start[0] * rowElems * es, size_t(count[0] * rowElems * es));
        std::vector<uint64_t> boxStart(zero), boxShape(ds.shape);
        boxStart[0] = start[0];
        boxShape[0] = count[0];
        copyBox(bytes.data(), boxStart, boxShape);
    } else {
        std::fill(out, out + total, 0.0f);  // chunks that were never written read as the fill value
        uint64_t chunkBytes = es;
        for (uint64_t c : ds.chunkShape) chunkBytes *= c;
        if (ds.chunkShape.size() != rank) fail("dataset \"" + ds.name + "\": chunk rank does not match");
        for (const auto& r : h.chunkIndex(ds)) {
            bool touches = true;
            for (size_t d = 0; d < rank; d++)
                if (r.offset[d] >= start[d] + count[d] || r.offset[d] + ds.chunkShape[d] <= start[d]) touches = false;
            if (!touches) continue;
            copyBox(h.loadChunk(ds, r, chunkBytes).data(), r.offset, ds.chunkShape);
        }
    }
}

}  // namespace crfhost
