/*
 * popsift/popsift.h -- PopSift / SiftJob, the asynchronous front end of the drop-in API.
 * Replaces popsift.h:40-167.  Same calls and ownership rules:
 *   - enqueue() copies the image and returns at once; the caller owns the SiftJob*
 *   - SiftJob::get() blocks and returns a Features* the caller must delete
 *   - uninit() must be called before the PopSift object goes away
 * What differs underneath: no CUDA / Boost in this header; work is spread over
 * every visible MI355X (POPSIFT_DEVICES, default "all") with several extraction
 * contexts per GPU (POPSIFT_CONTEXTS_PER_DEVICE, default 4), each with its own
 * HIP stream and buffers, behind the C ABI of popsift_hip.h.  Jobs complete through
 * their own futures, so any number of PopSift objects may coexist in a process.
 */
#pragma once

#include <condition_variable>
#include <future>
#include <mutex>
#include <queue>
#include <thread>
#include <vector>

#include <sched.h> /* cpu_set_t: a worker's CPUs */

#include "features.h"
#include "sift_conf.h"
#include "sift_extremum.h"

struct popsift_hip_ctx;

class SiftJob {
    std::promise<popsift::FeaturesBase*> _p;
    std::future<popsift::FeaturesBase*>  _f;
    int            _w;
    int            _h;
    unsigned char* _imageData; /* this job's copy of the image, in a recycled block of the pinned pool (features.h) */
    bool           _is_float;
    bool           _pinned = false;

public:
    /** byte image, values 0..255 */
    SiftJob(int w, int h, const unsigned char* imageData);
    /** float image, values [0..1[ */
    SiftJob(int w, int h, const float* imageData);
    ~SiftJob();

    popsift::FeaturesHost* get();  // same as getHost()
    popsift::FeaturesBase* getBase();
    popsift::FeaturesHost* getHost();
    popsift::FeaturesDev*  getDev();

    /** fulfil the promise (called by the worker) */
    void setFeatures(popsift::FeaturesBase* f);

    int                  getWidth() const { return _w; }
    int                  getHeight() const { return _h; }
    bool                 isFloat() const { return _is_float; }
    const unsigned char* getImageData() const { return _imageData; }
    /** extension: the job's image copy is page-locked (the worker then uploads it without a staging copy) */
    bool                 isPinned() const { return _pinned; }
};

namespace popsift {
/* extension: a sysfs cpulist ("0-15,128-143") as a CPU set; returns the number of CPUs named.  Re-entrant. */
int parseCpuList(const char* list, cpu_set_t* set);
}  // namespace popsift

class PopSift {
public:
    enum ImageMode { ByteImages, FloatImages };

    PopSift(ImageMode imode = ByteImages);
    PopSift(const popsift::Config& config,
            popsift::Config::ProcessingMode mode = popsift::Config::ExtractingMode,
            ImageMode imode = ByteImages);
    ~PopSift();

    /** Supply the configuration after default construction.  Returns false once
     *  images have been processed (the pyramid geometry is then fixed). */
    bool configure(const popsift::Config& config, bool force = false);

    /** Drains the queue, joins the workers and frees all GPU contexts. */
    void uninit();

    /** byte image, values 0..255 */
    SiftJob* enqueue(int w, int h, const unsigned char* imageData);
    /** float image, values 0..1 */
    SiftJob* enqueue(int w, int h, const float* imageData);

    /** deprecated blocking interface */
    inline void uninit(int /*pipe*/) { uninit(); }
    inline bool init(int /*pipe*/, int w, int h)
    {
        _last_init_w = w;
        _last_init_h = h;
        return true;
    }
    inline popsift::FeaturesBase* execute(int /*pipe*/, const unsigned char* imageData)
    {
        SiftJob* j = enqueue(_last_init_w, _last_init_h, imageData);
        if (!j) return 0;
        popsift::FeaturesBase* f = j->getBase();
        delete j;
        return f;
    }

    /** number of extraction contexts (GPUs x contexts per GPU) serving this object */
    int getContextCount() const { return (int)_workers.size(); }

private:
    struct Worker {
        std::thread      thread;
        popsift_hip_ctx* ctx = nullptr;
        int              device = 0;
        struct Pod { /* pinned staging for the POD features of a download, one per image of a batch */
            void*  p = nullptr;
            size_t cap = 0;
        };
        std::vector<Pod> pods;
        int              numa_node = -1; /* of the GPU; -1 unknown */
        bool             bind = false;   /* cpus holds the node's CPUs within the process's own affinity mask */
        cpu_set_t        cpus;
    };

    void start_workers(int w, int h);
    void worker_loop(Worker* me);

    std::vector<Worker*>    _workers;
    std::queue<SiftJob*>    _queue;
    std::mutex              _mtx;
    std::condition_variable _cv;
    bool                    _started = false;
    bool                    _stopped = false;

    popsift::Config _config;
    popsift::Config _shadow_config;
    int             _last_init_w = 0;
    int             _last_init_h = 0;
    ImageMode       _image_mode;
    popsift::Config::ProcessingMode _proc_mode = popsift::Config::ExtractingMode;
};
