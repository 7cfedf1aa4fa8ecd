/*
 * popsift/common/device_prop.h -- device enumeration / selection helper of the drop-in API.
 * Replaces common/device_prop.h:13-27.  The class keeps its name and its `popsift::cuda`
 * namespace so that callers (the demo: main.cpp:300-302) compile unchanged; `popsift::hip` is an
 * alias.  No GPU runtime header is needed to include this file.
 *
 * set(n) selects the GPU for PopSift objects created afterwards, like cudaSetDevice(n) did for
 * the reference's single pipeline.  Without a call to set(), and without POPSIFT_DEVICES in the
 * environment, a PopSift object spreads its work over every visible GPU.
 */
#pragma once

#include <string>
#include <vector>

namespace popsift {
namespace cuda {

class device_prop_t {
public:
    struct Properties {
        std::string        name;
        int                major, minor;
        unsigned long long totalGlobalMem;
        unsigned long long sharedMemPerBlock;
        int                warpSize;
        int                maxThreadsPerBlock;
        int                maxThreadsPerMultiProcessor;
        int                maxThreadsDim[3];
        int                maxGridSize[3];
        int                multiProcessorCount;
        bool               concurrentKernels;
        bool               canMapHostMemory;
        bool               unifiedAddressing;
    };

    device_prop_t();
    ~device_prop_t();

    void print();
    void set(int n, bool print_choice = false);

    int               getDeviceCount() const { return _num_devices; }
    const Properties& getProperties(int n) const { return _properties[(size_t)n]; }

    /* device chosen by the last set() in this process, -1 if none (read by PopSift) */
    static int chosenDevice();

private:
    int                     _num_devices;
    std::vector<Properties> _properties;
};

}  // namespace cuda
namespace hip = cuda;
}  // namespace popsift
