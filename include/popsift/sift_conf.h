/*
 * popsift/sift_conf.h -- popsift::Config, the run-time parameter block of the
 * drop-in API.  Same public names, enum values and defaults as the reference
 * (sift_conf.h:28-310, sift_conf.cu:17-39) so that caller code compiles
 * unchanged; unlike the reference, constructing a Config does not touch a device.
 * `namespace popart` is provided as an alias (the reference README still uses it).
 */
#pragma once

#include <string>

#define MAX_OCTAVES 20
#define MAX_LEVELS 10

#if defined(_MSC_VER)
#define DEPRECATED(func) __declspec(deprecated) func
#else
#define DEPRECATED(func) func __attribute__((deprecated))
#endif

namespace popsift {

struct Config {
    Config();

    /* How the 1-D Gauss tables are built / applied.  This build implements
     * VLFeat_Compute (default) and OpenCV_Compute; selecting another mode is fatal
     * at PopSift::configure time. */
    enum GaussMode { VLFeat_Compute, VLFeat_Relative, VLFeat_Relative_All, OpenCV_Compute, Fixed9, Fixed15 };

    /* Extremum refinement flavour */
    enum SiftMode { PopSift, OpenCV, VLFeat, Default = PopSift };

    enum LogMode { None, All };

    enum ScalingMode { ScaleDirect, ScaleDefault };

    /* Descriptor sampling scheme (all five are implemented; IGrid shares NoTile's kernel) */
    enum DescMode { Loop, ILoop, Grid, IGrid, NoTile };

    /* RootSift = L1-inspired (default), Classic = L2 + 0.2 clamp */
    enum NormMode { RootSift, Classic };

    enum GridFilterMode { RandomScale, LargestScaleFirst, SmallestScaleFirst };

    /* ExtractingMode: features are downloaded to the host (FeaturesHost).
     * MatchingMode: features stay on the GPU (FeaturesDev, SiftJob::getDev()) for FeaturesDev::match. */
    enum ProcessingMode { ExtractingMode, MatchingMode };

    /* ---- setters (sift_conf.cu:51-258) -------------------------------------- */
    void setGaussMode(const std::string& m);
    void setGaussMode(GaussMode m) { _gauss_mode = m; }
    void setMode(SiftMode m) { _sift_mode = m; }
    void setLogMode(LogMode mode = All) { _log_mode = mode; }
    void setScalingMode(ScalingMode mode = ScaleDefault) { _scaling_mode = mode; }
    void setVerbose(bool on = true) { verbose = on; }
    void setDescMode(const std::string& byname);
    void setDescMode(DescMode mode = Loop) { _desc_mode = mode; }

    void setDownsampling(float v) { _upscale_factor = -v; }
    void setOctaves(int v) { octaves = v; }
    void setLevels(int v) { levels = v; }
    void setSigma(float v) { sigma = v; }
    void setEdgeLimit(float v) { _edge_limit = v; }
    void setThreshold(float v) { _threshold = v; }
    void setInitialBlur(float blur);
    void setPrintGaussTables() { _print_gauss_tables = true; }
    void setFilterMaxExtrema(int extrema) { _filter_max_extrema = extrema; }
    void setFilterGridSize(int sz) { _filter_grid_size = sz; }
    void setFilterSorting(const std::string& direction);
    void setFilterSorting(GridFilterMode m) { _grid_filter_mode = m; }
    /* declared but never defined in the reference (sift_conf.h:101-114); defined here */
    void setMaxExtrema(int extrema) { _max_extrema = extrema; }
    void setMaxExtreme(int m) { _max_extrema = m; }
    void setGaussGroup(int) {}
    int  getGaussGroup() const { return 1; }
    void setDPOrientation(bool) {}

    void setNormMode(NormMode m) { _normalization_mode = m; }
    void setNormMode(const std::string& m);
    DEPRECATED(void setUseRootSift(bool on));
    void setNormalizationMultiplier(int mul) { _normalization_multiplier = mul; }

    /* ---- getters ------------------------------------------------------------- */
    bool  hasInitialBlur() const { return _assume_initial_blur; }
    float getInitialBlur() const { return _initial_blur; }
    /* threshold * 0.5 * 255 / levels (sift_conf.cu:275-278) */
    float getPeakThreshold() const { return (_threshold * 0.5f * 255.0f / levels); }
    float getThreshold() const { return _threshold; } /* extension: the raw setThreshold() value */
    bool  ifPrintGaussTables() const { return _print_gauss_tables; }
    GaussMode getGaussMode() const { return _gauss_mode; }
    static GaussMode   getGaussModeDefault() { return VLFeat_Compute; }
    static const char* getGaussModeUsage();
    SiftMode getSiftMode() const { return _sift_mode; }
    LogMode  getLogMode() const { return _log_mode; }
    bool     getUseRootSift() const { return _normalization_mode == RootSift; }
    NormMode getNormMode(NormMode = RootSift) const { return _normalization_mode; }
    static NormMode    getNormModeDefault() { return RootSift; }
    static const char* getNormModeUsage();
    int   getNormalizationMultiplier() const { return _normalization_multiplier; }
    float getUpscaleFactor() const { return _upscale_factor; }
    int   getMaxExtrema() const { return _max_extrema; }
    bool  getCanFilterExtrema() const { return true; } /* sift_conf.h:183: false only for CUDA < 8 builds */
    int   getFilterMaxExtrema() const { return _filter_max_extrema; }
    int   getFilterGridSize() const { return _filter_grid_size; }
    GridFilterMode getFilterSorting() const { return _grid_filter_mode; }
    ScalingMode    getScalingMode() const { return _scaling_mode; }
    DescMode       getDescMode() const { return _desc_mode; }

    /* compares the 14 fields that decide whether tables must be rebuilt (sift_conf.cu:285-303) */
    bool equal(const Config& other) const;

    /* ---- public data members, as in the reference ------------------------------ */
    int   octaves;     /* < 0: floor(log2(min(w,h))) - 3 + 2^upscale, decided by the first image */
    int   levels;      /* DoG levels searched per octave (Gaussian planes = levels + 3) */
    float sigma;
    float _edge_limit;
    bool  verbose;

private:
    float          _threshold;
    float          _upscale_factor;
    LogMode        _log_mode;
    ScalingMode    _scaling_mode;
    DescMode       _desc_mode;
    GridFilterMode _grid_filter_mode;
    int            _max_extrema;
    int            _filter_max_extrema;
    int            _filter_grid_size;
    GaussMode      _gauss_mode;
    SiftMode       _sift_mode;
    bool           _assume_initial_blur;
    float          _initial_blur;
    NormMode       _normalization_mode;
    int            _normalization_multiplier;
    bool           _print_gauss_tables;
};

inline bool operator==(const Config& l, const Config& r) { return l.equal(r); }
inline bool operator!=(const Config& l, const Config& r) { return !l.equal(r); }

}  // namespace popsift

namespace popart = popsift;
