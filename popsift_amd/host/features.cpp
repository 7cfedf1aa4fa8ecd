/* Feature / FeaturesHost (replaces features.cu:23-122,308-334). */
#include "popsift/features.h"

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <iomanip>
#include <map>
#include <mutex>
#include <unistd.h>

#include "popsift_hip.h"

namespace popsift {

namespace {

/*
 * Result buffers come from a process-wide pool of pinned host blocks (power-of-two size classes):
 * the device-to-host copy of a job then runs at PCIe speed straight into the memory the caller
 * sees.  The reference pins and unpins the freshly allocated arrays around every download
 * (cudaHostRegister, features.cu:84-109), which costs milliseconds per image.  When pinned
 * memory cannot be had (no GPU runtime), blocks are plain page-aligned allocations.
 */
class BlockPool {
    struct Block {
        void*  p;
        size_t cap;
        bool   pinned;
    };
    std::mutex                   _m;
    std::multimap<size_t, Block> _free; /* by capacity */
    std::map<void*, Block>       _live;
    static constexpr size_t      MAX_CACHED = (size_t)2 << 30; /* bytes kept for reuse */
    size_t                       _cached = 0;

    static size_t round_up(size_t n)
    {
        size_t c = 4096;
        while (c < n) c <<= 1;
        return c;
    }
    static void release(const Block& b)
    {
        if (b.pinned) popsift_hip_host_free(b.p);
        else free(b.p);
    }

public:
    void* get(size_t bytes)
    {
        const size_t                cap = round_up(bytes);
        std::lock_guard<std::mutex> lk(_m);
        auto                        it = _free.lower_bound(cap);
        if (it != _free.end() && it->first <= 2 * cap) {
            Block b = it->second;
            _free.erase(it);
            _cached -= b.cap;
            _live[b.p] = b;
            return b.p;
        }
        Block b{popsift_hip_host_alloc(cap), cap, true};
        if (!b.p) {
            b.pinned = false;
            if (posix_memalign(&b.p, (size_t)sysconf(_SC_PAGESIZE), cap) != 0) return 0;
        }
        _live[b.p] = b;
        return b.p;
    }
    void put(void* p)
    {
        if (!p) return;
        std::lock_guard<std::mutex> lk(_m);
        auto                        it = _live.find(p);
        if (it == _live.end()) return;
        Block b = it->second;
        _live.erase(it);
        if (_cached + b.cap > MAX_CACHED) {
            release(b);
        } else {
            _cached += b.cap;
            _free.insert(std::make_pair(b.cap, b));
        }
    }
};

BlockPool& pool()
{
    static BlockPool* p = new BlockPool; /* never destroyed: results may outlive static teardown */
    return *p;
}

}  // namespace

FeaturesHost::FeaturesHost() : _ext(0), _ori(0) {}

FeaturesHost::FeaturesHost(int num_ext, int num_ori) : _ext(0), _ori(0) { reset(num_ext, num_ori); }

FeaturesHost::~FeaturesHost()
{
    pool().put(_ext);
    pool().put(_ori);
}

void FeaturesHost::reset(int num_ext, int num_ori)
{
    pool().put(_ext);
    pool().put(_ori);
    /* page-aligned like the reference (features.cu:63,72); zero-sized results stay valid objects */
    _ext = (Feature*)pool().get(std::max<size_t>((size_t)num_ext * sizeof(Feature), 1));
    if (_ext == 0) {
        std::cerr << __FILE__ << ":" << __LINE__ << " Runtime error:" << std::endl
                  << "    Failed to (re)allocate memory for downloading " << num_ext << " features" << std::endl;
        exit(-1);
    }
    _ori = (Descriptor*)pool().get(std::max<size_t>((size_t)num_ori * sizeof(Descriptor), 1));
    if (_ori == 0) {
        std::cerr << __FILE__ << ":" << __LINE__ << " Runtime error:" << std::endl
                  << "    Failed to (re)allocate memory for downloading " << num_ori << " descriptors" << std::endl;
        exit(-1);
    }
    setFeatureCount(num_ext);
    setDescriptorCount(num_ori);
}

void FeaturesHost::print(std::ostream& ostr, bool write_as_uchar) const
{
    for (int i = 0; i < size(); i++) _ext[i].print(ostr, write_as_uchar);
}

std::ostream& operator<<(std::ostream& ostr, const FeaturesHost& feature)
{
    feature.print(ostr, false);
    return ostr;
}

/* text format of features.cu:308-328 */
void Feature::print(std::ostream& ostr, bool write_as_uchar) const
{
    const float sigval = 1.0f / (sigma * sigma);
    for (int ori = 0; ori < num_ori; ori++) {
        ostr << xpos << " " << ypos << " " << sigval << " 0 " << sigval << " ";
        if (write_as_uchar) {
            for (int i = 0; i < 128; i++) ostr << roundf(desc[ori]->features[i]) << " ";
        } else {
            ostr << std::setprecision(3);
            for (int i = 0; i < 128; i++) ostr << desc[ori]->features[i] << " ";
            ostr << std::setprecision(6);
        }
        ostr << std::endl;
    }
}

std::ostream& operator<<(std::ostream& ostr, const Feature& feature)
{
    feature.print(ostr, false);
    return ostr;
}

}  // namespace popsift
