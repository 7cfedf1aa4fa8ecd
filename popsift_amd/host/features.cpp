/* Feature / FeaturesHost (replaces features.cu:23-122,308-334). */
#include "popsift/features.h"

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <iomanip>
#include <unistd.h>

namespace popsift {

FeaturesHost::FeaturesHost() : _ext(0), _ori(0) {}

FeaturesHost::FeaturesHost(int num_ext, int num_ori) : _ext(0), _ori(0) { reset(num_ext, num_ori); }

FeaturesHost::~FeaturesHost()
{
    free(_ext);
    free(_ori);
}

void FeaturesHost::reset(int num_ext, int num_ori)
{
    free(_ext);
    free(_ori);
    _ext = 0;
    _ori = 0;
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    void*        p = 0;
    /* page-aligned like the reference (features.cu:63,72); zero-sized results stay valid objects */
    if (posix_memalign(&p, page, std::max<size_t>((size_t)num_ext * sizeof(Feature), page)) != 0) {
        std::cerr << __FILE__ << ":" << __LINE__ << " Runtime error:" << std::endl
                  << "    Failed to (re)allocate memory for downloading " << num_ext << " features" << std::endl;
        exit(-1);
    }
    _ext = (Feature*)p;
    if (posix_memalign(&p, page, std::max<size_t>((size_t)num_ori * sizeof(Descriptor), page)) != 0) {
        std::cerr << __FILE__ << ":" << __LINE__ << " Runtime error:" << std::endl
                  << "    Failed to (re)allocate memory for downloading " << num_ori << " descriptors" << std::endl;
        exit(-1);
    }
    _ori = (Descriptor*)p;
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
