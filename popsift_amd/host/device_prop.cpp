/* device_prop_t over the C ABI (replaces common/device_prop.cu:17-95). */
#include "popsift/common/device_prop.h"

#include <atomic>
#include <cstdlib>
#include <iostream>

#include "popsift_hip.h"

namespace popsift {
namespace cuda {

using namespace std;

namespace {
std::atomic<int> g_chosen(-1);

[[noreturn]] void fatal(const std::string& msg)
{
    /* POP_CUDA_FATAL_TEST convention (debug_macros.h:141-146): message, then exit */
    cerr << __FILE__ << endl << "E    " << msg << endl;
    exit(-1);
}
}  // namespace

device_prop_t::device_prop_t() : _num_devices(0)
{
    if (popsift_hip_device_count(&_num_devices) != POPSIFT_HIP_OK) fatal("Cannot count devices");
    for (int n = 0; n < _num_devices; n++) {
        popsift_hip_device_info di;
        if (popsift_hip_get_device_info(n, &di) != POPSIFT_HIP_OK) fatal("Cannot get properties for a device");
        Properties p;
        p.name = di.name;
        p.major = di.arch_major;
        p.minor = di.arch_minor;
        p.totalGlobalMem = di.total_mem;
        p.sharedMemPerBlock = di.lds_per_block;
        p.warpSize = di.wave_size;
        p.maxThreadsPerBlock = di.max_threads_per_block;
        p.maxThreadsPerMultiProcessor = di.max_threads_per_cu;
        for (int i = 0; i < 3; i++) {
            p.maxThreadsDim[i] = di.max_block[i];
            p.maxGridSize[i] = di.max_grid[i];
        }
        p.multiProcessorCount = di.cu_count;
        p.concurrentKernels = di.concurrent_kernels != 0;
        p.canMapHostMemory = di.can_map_host != 0;
        p.unifiedAddressing = di.unified_addressing != 0;
        _properties.push_back(p);
    }
}

device_prop_t::~device_prop_t() {}

/* same lines and labels as device_prop.cu:39-70 (the labels keep the reference's vocabulary) */
void device_prop_t::print()
{
    for (const Properties& p : _properties) {
        std::cout << "Device information:" << endl
                  << "    Name: " << p.name << endl
                  << "    Compute Capability:    " << p.major << "." << p.minor << endl
                  << "    Total device mem:      " << p.totalGlobalMem << " B " << p.totalGlobalMem / 1024 << " kB "
                  << p.totalGlobalMem / (1024 * 1024) << " MB " << endl
                  << "    Per-block shared mem:  " << p.sharedMemPerBlock << endl
                  << "    Warp size:             " << p.warpSize << endl
                  << "    Max threads per block: " << p.maxThreadsPerBlock << endl
                  << "    Max threads per SM(X): " << p.maxThreadsPerMultiProcessor << endl
                  << "    Max block sizes:       "
                  << "{" << p.maxThreadsDim[0] << "," << p.maxThreadsDim[1] << "," << p.maxThreadsDim[2] << "}" << endl
                  << "    Max grid sizes:        "
                  << "{" << p.maxGridSize[0] << "," << p.maxGridSize[1] << "," << p.maxGridSize[2] << "}" << endl
                  << "    Number of SM(x)s:      " << p.multiProcessorCount << endl
                  << "    Concurrent kernels:    " << (p.concurrentKernels ? "yes" : "no") << endl
                  << "    Mapping host memory:   " << (p.canMapHostMemory ? "yes" : "no") << endl
                  << "    Unified addressing:    " << (p.unifiedAddressing ? "yes" : "no") << endl
                  << endl;
    }
}

void device_prop_t::set(int n, bool print_choice)
{
    if (n < 0 || n >= _num_devices) fatal("Cannot set device " + std::to_string(n));
    g_chosen.store(n);
    if (print_choice) std::cout << "Choosing device " << n << ": " << _properties[(size_t)n].name << std::endl;
}

int device_prop_t::chosenDevice() { return g_chosen.load(); }

}  // namespace cuda
}  // namespace popsift
