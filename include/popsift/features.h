/*
 * popsift/features.h -- result containers of the drop-in API.
 * Replaces features.h:22-96 (Feature, FeaturesBase, FeaturesHost, typedef Features).
 * Layouts are the reference's: Feature keeps its four Descriptor* which point
 * into the FeaturesHost's own descriptor array.  FeaturesDev (features.h:98-118)
 * holds the results of an image in GPU memory and matches two such sets.
 */
#pragma once

#include <iostream>
#include <vector>

#include "sift_constants.h"
#include "sift_extremum.h"

struct popsift_hip_devfeatures;

namespace popsift {

struct Feature {
    int         debug_octave;
    float       xpos;
    float       ypos;
    float       sigma;
    int         num_ori;
    float       orientation[ORIENTATION_MAX_COUNT];
    Descriptor* desc[ORIENTATION_MAX_COUNT];

    void print(std::ostream& ostr, bool write_as_uchar) const;
};

std::ostream& operator<<(std::ostream& ostr, const Feature& feature);

class FeaturesBase {
    int _num_ext;
    int _num_ori;

public:
    FeaturesBase() : _num_ext(0), _num_ori(0) {}
    virtual ~FeaturesBase() {}

    inline int size() const { return _num_ext; }
    inline int getFeatureCount() const { return _num_ext; }
    inline int getDescriptorCount() const { return _num_ori; }

    inline void setFeatureCount(int num_ext) { _num_ext = num_ext; }
    inline void setDescriptorCount(int num_ori) { _num_ori = num_ori; }
};

class FeaturesHost : public FeaturesBase {
    Feature*    _ext;
    Descriptor* _ori;

public:
    FeaturesHost();
    FeaturesHost(int num_ext, int num_ori);
    virtual ~FeaturesHost();

    typedef Feature*       F_iterator;
    typedef const Feature* F_const_iterator;

    inline F_iterator       begin() { return _ext; }
    inline F_const_iterator begin() const { return _ext; }
    inline F_iterator       end() { return &_ext[size()]; }
    inline F_const_iterator end() const { return &_ext[size()]; }

    /* (re)allocates page-aligned arrays for num_ext features / num_ori descriptors */
    void reset(int num_ext, int num_ori);
    /* host-memory registration with the GPU runtime; no-ops here (the arrays already are
     * pinned blocks from a process-wide pool) but kept for source compatibility */
    void pin() {}
    void unpin() {}

    inline Feature*    getFeatures() { return _ext; }
    inline Descriptor* getDescriptors() { return _ori; }

    /* one line per (feature, orientation): x y 1/s^2 0 1/s^2 d0 .. d127 */
    void print(std::ostream& ostr, bool write_as_uchar) const;
};

typedef FeaturesHost Features;

/* Extensions: the result arrays of FeaturesHost are pinned blocks from a process-wide pool; free blocks are cached for
 * reuse (POPSIFT_PINNED_CACHE_MB, default 2048) and released when the last PopSift object of the process is shut down.
 * releasePinnedCache() releases them at once; pinnedCacheBytes() tells how much is cached. */
void   releasePinnedCache();
size_t pinnedCacheBytes();
/* the NUMA node (or -1) whose free list serves the calling thread's result blocks; PopSift's workers set it */
void   setPinnedPoolNode(int node);
/* A block of the same pool (SiftJob keeps its copy of the caller's image in one: no allocation, no page faults and no
 * second staging copy per job).  *pinned tells whether the block is page-locked (it is not when no GPU runtime is there). */
void*  pinnedBlockGet(size_t bytes, bool* pinned);
void   pinnedBlockPut(void* block);

std::ostream& operator<<(std::ostream& ostr, const FeaturesHost& feature);

/*
 * Device-resident results of one image (PopSift in Config::MatchingMode, SiftJob::getDev()).
 * The three arrays live in the memory of the GPU that extracted the image: Feature records whose
 * desc[] point into the descriptor array, the descriptors, and the descriptor -> feature map.
 */
class FeaturesDev : public FeaturesBase {
    popsift_hip_devfeatures* _set;

public:
    FeaturesDev();
    FeaturesDev(int num_ext, int num_ori);
    /* takes ownership of a set made by the C ABI (popsift_hip_clone_results) */
    explicit FeaturesDev(popsift_hip_devfeatures* adopt);
    virtual ~FeaturesDev();

    void reset(int num_ext, int num_ori);

    /* brute-force 2-nearest-neighbour search of every descriptor of this set among the descriptors of
     * `other`; prints one "accept ..." / "reject ..." line per descriptor on stdout, as the reference does */
    void match(FeaturesDev* other);

    /* extension: the same search, results returned instead of printed */
    struct Match {
        int   best, second; /* descriptor indices in `other` */
        bool  accept;       /* dist_best / dist_second < 0.8 */
        float dist_best, dist_second; /* squared L2 */
    };
    std::vector<Match> matchAndGet(FeaturesDev* other);

    /* DEVICE pointers */
    Feature*    getFeatures();
    Descriptor* getDescriptors();
    int*        getReverseMap();

    int                      getDevice() const;
    popsift_hip_devfeatures* getHandle() { return _set; }
};

}  // namespace popsift
