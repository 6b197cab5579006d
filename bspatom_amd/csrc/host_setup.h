// host_setup.h -- host-side problem set-up (see host_setup.cpp)
#pragma once
#include <vector>
#include "../../include/bspatom.h"

namespace bsp {

struct HostSetup {
    bspatom_input in;
    int nfun, k, ka, nkp, nointv, nbc1, nbc2, lmax, nintv_exp, nintv_lin;
    double gsize;
    int numn[3], ntot;
    double alphan[3], bl[4];
    std::vector<double> rt, aind, xg, wg, vpot;
};

void input_defaults(bspatom_input *in);
int derive(const bspatom_input &in, HostSetup *h);
void gauleg(double x1, double x2, double *x, double *w, int n);
void build_grid(HostSetup *h);
double selpot(const HostSetup &h, double r);
void build_vpot(HostSetup *h);

}  // namespace bsp
