/* Feature / FeaturesHost (replaces features.cu:23-122,308-334). */
#include "popsift/features.h"

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iomanip>
#include <map>
#include <mutex>
#include <unistd.h>

#include "popsift_hip.h"

namespace popsift {

namespace {

/*
 * Result buffers come from a process-wide pool of pinned host blocks: the device-to-host copy of a job then runs at
 * PCIe speed straight into the memory the caller sees.  The reference pins and unpins the freshly allocated arrays
 * around every download (cudaHostRegister, features.cu:84-109), which costs milliseconds per image.  When pinned
 * memory cannot be had (no GPU runtime), blocks are plain page-aligned allocations.
 *   - size classes grow by a quarter (a 70 MB result takes an 82 MB block, not a 128 MB power of two);
 *   - blocks are allocated and released OUTSIDE the pool lock (pinning tens of MB takes tens of milliseconds);
 *   - at most POPSIFT_PINNED_CACHE_MB (default 2 GiB) of free blocks are kept, and none once the last PopSift object
 *     of the process has been shut down (trim(), called from PopSift::uninit): a long-running host application does
 *     not keep gigabytes of unswappable memory after it is done extracting.
 */
class BlockPool {
    struct Block {
        void*  p;
        size_t cap;
        bool   pinned;
        int    node; /* NUMA node of the thread that allocated (first touched) it, -1 unknown */
    };
    typedef std::pair<int, size_t> Key; /* (node, capacity): a worker reuses blocks of its own node only */
    std::mutex                _m;
    std::multimap<Key, Block> _free;
    std::map<void*, Block>       _live;
    const size_t                 MAX_CACHED = cache_limit();
    size_t                       _cached = 0;

    static size_t cache_limit()
    {
        const char* e = getenv("POPSIFT_PINNED_CACHE_MB");
        const long  mb = e ? atol(e) : 2048;
        return (size_t)(mb < 0 ? 0 : mb) << 20;
    }
    static size_t round_up(size_t n)
    {
        size_t c = 4096;
        while (c < n) c = (c + c / 4 + 4095) & ~(size_t)4095;
        return c;
    }
    static void release(const Block& b)
    {
        if (b.pinned) popsift_hip_host_free(b.p);
        else free(b.p);
    }

public:
    void* get(size_t bytes, int node, bool* pinned = 0)
    {
        const size_t cap = round_up(bytes);
        {
            std::lock_guard<std::mutex> lk(_m);
            auto                        it = _free.lower_bound(Key(node, cap));
            if (it != _free.end() && it->first.first == node && it->first.second <= cap + cap / 2) {
                Block b = it->second;
                _free.erase(it);
                _cached -= b.cap;
                _live[b.p] = b;
                if (pinned) *pinned = b.pinned;
                return b.p;
            }
        }
        Block b{popsift_hip_host_alloc(cap), cap, true, node}; /* not under the lock: other workers keep going */
        if (!b.p) {
            b.pinned = false;
            if (posix_memalign(&b.p, (size_t)sysconf(_SC_PAGESIZE), cap) != 0) return 0;
        }
        std::lock_guard<std::mutex> lk(_m);
        _live[b.p] = b;
        if (pinned) *pinned = b.pinned;
        return b.p;
    }
    void put(void* p)
    {
        if (!p) return;
        Block b;
        {
            std::lock_guard<std::mutex> lk(_m);
            auto                        it = _live.find(p);
            if (it == _live.end()) return;
            b = it->second;
            _live.erase(it);
            if (_cached + b.cap <= MAX_CACHED) {
                _cached += b.cap;
                _free.insert(std::make_pair(Key(b.node, b.cap), b));
                return;
            }
        }
        release(b);
    }
    /* give every cached block back to the system (blocks held by live results are untouched) */
    void trim()
    {
        std::multimap<Key, Block> drop;
        {
            std::lock_guard<std::mutex> lk(_m);
            drop.swap(_free);
            _cached = 0;
        }
        for (auto& kv : drop) release(kv.second);
    }
    size_t cached_bytes()
    {
        std::lock_guard<std::mutex> lk(_m);
        return _cached;
    }
};

BlockPool& pool()
{
    static BlockPool* p = new BlockPool; /* never destroyed: results may outlive static teardown */
    return *p;
}

}  // namespace

thread_local int t_pool_node = -1;

void   setPinnedPoolNode(int node) { t_pool_node = node; }
void   releasePinnedCache() { pool().trim(); }
size_t pinnedCacheBytes() { return pool().cached_bytes(); }
void*  pinnedBlockGet(size_t bytes, bool* pinned) { return pool().get(std::max<size_t>(bytes, 1), t_pool_node, pinned); }
void   pinnedBlockPut(void* block) { pool().put(block); }

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
    _ext = (Feature*)pool().get(std::max<size_t>((size_t)num_ext * sizeof(Feature), 1), t_pool_node);
    if (_ext == 0) {
        std::cerr << __FILE__ << ":" << __LINE__ << " Runtime error:" << std::endl
                  << "    Failed to (re)allocate memory for downloading " << num_ext << " features" << std::endl;
        exit(-1);
    }
    _ori = (Descriptor*)pool().get(std::max<size_t>((size_t)num_ori * sizeof(Descriptor), 1), t_pool_node);
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

/* ------------------------------------------------------------------------ FeaturesDev */

namespace {
[[noreturn]] void dev_fatal(const char* what, int rc)
{
    std::cerr << __FILE__ << std::endl << "E    " << what << ": " << popsift_hip_strerror(rc) << std::endl;
    exit(-1);
}
}  // namespace

FeaturesDev::FeaturesDev() : _set(0) {}

FeaturesDev::FeaturesDev(int num_ext, int num_ori) : _set(0) { reset(num_ext, num_ori); }

FeaturesDev::FeaturesDev(popsift_hip_devfeatures* adopt) : _set(adopt)
{
    int nf = 0, nd = 0;
    if (_set) popsift_hip_devfeatures_info(_set, 0, &nf, &nd);
    setFeatureCount(nf);
    setDescriptorCount(nd);
}

FeaturesDev::~FeaturesDev() { popsift_hip_devfeatures_free(_set); }

/* features.cu:148-160 */
void FeaturesDev::reset(int num_ext, int num_ori)
{
    int dev = 0;
    if (_set) popsift_hip_devfeatures_info(_set, &dev, 0, 0);
    popsift_hip_devfeatures_free(_set);
    _set = 0;
    const int rc = popsift_hip_devfeatures_alloc(dev, num_ext, num_ori, &_set);
    if (rc != POPSIFT_HIP_OK) dev_fatal("cannot allocate device feature arrays", rc);
    setFeatureCount(num_ext);
    setDescriptorCount(num_ori);
}

Feature* FeaturesDev::getFeatures()
{
    void* p = 0;
    if (_set) popsift_hip_devfeatures_ptrs(_set, &p, 0, 0);
    return (Feature*)p;
}
Descriptor* FeaturesDev::getDescriptors()
{
    void* p = 0;
    if (_set) popsift_hip_devfeatures_ptrs(_set, 0, &p, 0);
    return (Descriptor*)p;
}
int* FeaturesDev::getReverseMap()
{
    void* p = 0;
    if (_set) popsift_hip_devfeatures_ptrs(_set, 0, 0, &p);
    return (int*)p;
}
int FeaturesDev::getDevice() const
{
    int dev = 0;
    if (_set) popsift_hip_devfeatures_info(_set, &dev, 0, 0);
    return dev;
}

std::vector<FeaturesDev::Match> FeaturesDev::matchAndGet(FeaturesDev* other)
{
    std::vector<Match> res;
    if (!_set || !other || !other->_set) return res;
    const int                      l_len = getDescriptorCount();
    std::vector<popsift_hip_match> raw((size_t)l_len);
    const int                      rc = popsift_hip_match_sets(_set, other->_set, raw.data());
    if (rc != POPSIFT_HIP_OK) dev_fatal("matching failed", rc);
    res.resize((size_t)l_len);
    for (int i = 0; i < l_len; i++) {
        const popsift_hip_match& m = raw[(size_t)i];
        res[(size_t)i] = Match{m.best, m.second, m.accept != 0, m.dist_best, m.dist_second};
    }
    return res;
}

/* FeaturesDev::match + show_distance (features.cu:222-300): the reference prints from the device */
void FeaturesDev::match(FeaturesDev* other)
{
    const std::vector<Match> m = matchAndGet(other);
    const int                l_len = getDescriptorCount();
    const int                r_len = other ? other->getDescriptorCount() : 0;
    if (l_len == 0 || r_len == 0) return;
    std::vector<int> l_fem((size_t)l_len), r_fem((size_t)r_len);
    popsift_hip_devfeatures_download(_set, 0, l_fem.data());
    popsift_hip_devfeatures_download(other->_set, 0, r_fem.data());
    for (int i = 0; i < l_len; i++) {
        const Match& x = m[(size_t)i];
        printf("%s feat %4d [%4d] matches feat %4d [%4d] ( 2nd feat %4d [%4d] ) dist %.3f vs %.3f\n",
               x.accept ? "accept" : "reject", l_fem[(size_t)i], i, r_fem[(size_t)x.best], x.best,
               r_fem[(size_t)x.second], x.second, x.dist_best, x.dist_second);
    }
}

}  // namespace popsift
