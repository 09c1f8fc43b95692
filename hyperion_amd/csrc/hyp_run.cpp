// hyp_run.cpp -- native counterpart of the reference's `hyperion_car` / `hyperion_sph` / ... executables:
//
//     hyperion_<grid> [-f] input.rtin output.rtout
//
// reads the HDF5 input the reference's Python front-end writes (hyperion/model/model.py:486-760), runs the iteration
// sequence of `program main` (src/main/main.f90:74-344) on the HIP engine through the C ABI of
// include/hyperion_amd.h, and writes the .rtout the reference's ModelOutput reads.  No Python, no h5py: libhdf5 (C API)
// and libhyperion_amd.so only, so the reference's unmodified scripts/hyperion:92 finds a drop-in on PATH.
// Under the `_mpi` names (hyperion_car_mpi, ...: what scripts/hyperion:62-92 starts with `mpirun -n N`) or whenever the
// launcher's environment names a rank (RANK / OMPI_COMM_WORLD_RANK / PMI_RANK / SLURM_PROCID), the process is one rank of
// N, one GPU each: packets are split by id range, ONE ncclAllReduce (RCCL over xGMI, rccl.h) of the accumulator block per
// iteration replaces mp_collect_physical_arrays / mp_collect_images (src/mpi/mpi_routines.f90:272-361), every rank runs the
// epilogue, rank 0 alone writes the output.  `--ranks N` starts the N ranks itself (no MPI installation needed).
//
// What is read, and the defaults, follow src/main/setup_rt.f90:38-302 (root attributes, /Output),
// src/grid/grid_geometry_*.f90 (geometry), src/dust/dust_type_4elem.f90:78-293 (dust), src/sources/source_type.f90:102-322
// (sources), src/images/image_type.f90:153-335 + images_peeled.f90:306-345 + images_binned.f90:42-56 (image groups);
// what is written follows main.f90:130-344, src/grid/grid_generic.f90:29-130 and image_type.f90:608-788.
// Failure convention of the reference: message on stderr, the output lacks `date_ended`, non-zero exit status.
#include <hdf5.h>
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <sys/wait.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/hyperion_amd.h"

namespace {

const double C_CGS = 29979245800.0;
const char *FORTRAN_VERSION = "1.0.0";      // src/main/main.f90 `fortran_version`

struct Fail : std::runtime_error { using std::runtime_error::runtime_error; };

std::string fmt(const char *f, ...)
{
    char buf[1024];
    va_list ap; va_start(ap, f); vsnprintf(buf, sizeof buf, f, ap); va_end(ap);
    return buf;
}

// ---------------------------------------------------------------------------------------------------------------
// HDF5 helpers
// ---------------------------------------------------------------------------------------------------------------
struct Hid {        // closes what it holds
    hid_t id; int kind;     // 0 file, 1 group, 2 dataset, 3 attribute, 4 type, 5 space, 6 plist
    Hid(hid_t i, int k) : id(i), kind(k) {}
    Hid(const Hid &) = delete;
    ~Hid()
    {
        if (id < 0) return;
        switch (kind) {
        case 0: H5Fclose(id); break; case 1: H5Gclose(id); break; case 2: H5Dclose(id); break;
        case 3: H5Aclose(id); break; case 4: H5Tclose(id); break; case 5: H5Sclose(id); break; default: H5Pclose(id);
        }
    }
    operator hid_t() const { return id; }
};

bool has_attr(hid_t obj, const char *name) { return H5Aexists(obj, name) > 0; }
bool has_link(hid_t loc, const char *name) { return H5Lexists(loc, name, H5P_DEFAULT) > 0; }

std::string trim(std::string s)
{
    while (!s.empty() && (s.back() == ' ' || s.back() == '\0' || s.back() == '\n')) s.pop_back();
    size_t i = 0;
    while (i < s.size() && s[i] == ' ') i++;
    return s.substr(i);
}

std::string attr_str(hid_t obj, const char *name)
{
    if (!has_attr(obj, name)) throw Fail(fmt("attribute %s does not exist", name));
    Hid a(H5Aopen(obj, name, H5P_DEFAULT), 3);
    Hid t(H5Aget_type(a), 4);
    if (H5Tget_class(t) != H5T_STRING) throw Fail(fmt("attribute %s is not a string", name));
    if (H5Tis_variable_str(t) > 0) {
        char *p = nullptr;
        Hid mt(H5Tcopy(H5T_C_S1), 4);
        H5Tset_size(mt, H5T_VARIABLE);
        H5Tset_cset(mt, H5Tget_cset(t));
        if (H5Aread(a, mt, &p) < 0) throw Fail(fmt("cannot read attribute %s", name));
        std::string s = p ? p : "";
        if (p) H5free_memory(p);
        return trim(s);
    }
    size_t n = H5Tget_size(t);
    std::vector<char> buf(n + 1, 0);
    if (H5Aread(a, t, buf.data()) < 0) throw Fail(fmt("cannot read attribute %s", name));
    return trim(std::string(buf.data(), strnlen(buf.data(), n)));
}

double attr_dbl(hid_t obj, const char *name)
{
    if (!has_attr(obj, name)) throw Fail(fmt("attribute %s does not exist", name));
    Hid a(H5Aopen(obj, name, H5P_DEFAULT), 3);
    double v = 0.0;
    if (H5Aread(a, H5T_NATIVE_DOUBLE, &v) < 0) throw Fail(fmt("cannot read attribute %s", name));
    return v;
}

long long attr_int(hid_t obj, const char *name)
{
    if (!has_attr(obj, name)) throw Fail(fmt("attribute %s does not exist", name));
    Hid a(H5Aopen(obj, name, H5P_DEFAULT), 3);
    long long v = 0;
    if (H5Aread(a, H5T_NATIVE_LLONG, &v) < 0) throw Fail(fmt("cannot read attribute %s", name));
    return v;
}

std::vector<double> attr_dbl_array(hid_t obj, const char *name)
{
    Hid a(H5Aopen(obj, name, H5P_DEFAULT), 3);
    Hid s(H5Aget_space(a), 5);
    hssize_t n = H5Sget_simple_extent_npoints(s);
    std::vector<double> v((size_t)std::max<hssize_t>(n, 1));
    if (H5Aread(a, H5T_NATIVE_DOUBLE, v.data()) < 0) throw Fail(fmt("cannot read attribute %s", name));
    return v;
}

bool attr_bool(hid_t obj, const char *name)      // the reference writes 'yes' / 'no'
{
    std::string s = attr_str(obj, name);
    std::transform(s.begin(), s.end(), s.begin(), ::tolower);
    if (s == "yes" || s == "y" || s == "true") return true;
    if (s == "no" || s == "n" || s == "false") return false;
    throw Fail(fmt("cannot interpret attribute %s = '%s' as a boolean", name, s.c_str()));
}

std::vector<std::string> children(hid_t loc)
{
    std::vector<std::string> out;
    H5G_info_t info;
    if (H5Gget_info(loc, &info) < 0) throw Fail("cannot list group");
    for (hsize_t i = 0; i < info.nlinks; i++) {
        ssize_t n = H5Lget_name_by_idx(loc, ".", H5_INDEX_NAME, H5_ITER_INC, i, nullptr, 0, H5P_DEFAULT);
        std::string s((size_t)n + 1, '\0');
        H5Lget_name_by_idx(loc, ".", H5_INDEX_NAME, H5_ITER_INC, i, &s[0], (size_t)n + 1, H5P_DEFAULT);
        s.resize((size_t)n);
        out.push_back(s);
    }
    std::sort(out.begin(), out.end());
    return out;
}

bool is_group(hid_t loc, const std::string &name)
{
    H5O_info_t oi;
    if (H5Oget_info_by_name(loc, name.c_str(), &oi, H5P_DEFAULT) < 0) return false;
    return oi.type == H5O_TYPE_GROUP;
}

// a whole numeric dataset as doubles, with its shape
std::vector<double> read_doubles(hid_t loc, const char *name, std::vector<hsize_t> *dims = nullptr)
{
    if (!has_link(loc, name)) throw Fail(fmt("dataset %s does not exist", name));
    Hid d(H5Dopen2(loc, name, H5P_DEFAULT), 2);
    if (d < 0) throw Fail(fmt("cannot open dataset %s", name));
    Hid s(H5Dget_space(d), 5);
    int nd = H5Sget_simple_extent_ndims(s);
    std::vector<hsize_t> dm((size_t)std::max(nd, 0));
    if (nd > 0) H5Sget_simple_extent_dims(s, dm.data(), nullptr);
    size_t n = 1;
    for (hsize_t x : dm) n *= (size_t)x;
    std::vector<double> v(n);
    if (n && H5Dread(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, v.data()) < 0) throw Fail(fmt("cannot read dataset %s", name));
    if (dims) *dims = dm;
    return v;
}

std::vector<int32_t> read_ints(hid_t loc, const char *name)
{
    std::vector<double> v = read_doubles(loc, name);
    std::vector<int32_t> o(v.size());
    for (size_t i = 0; i < v.size(); i++) o[i] = (int32_t)v[i];
    return o;
}

bool table_has(hid_t loc, const char *dset, const char *field)
{
    Hid d(H5Dopen2(loc, dset, H5P_DEFAULT), 2);
    if (d < 0) return false;
    Hid t(H5Dget_type(d), 4);
    return H5Tget_class(t) == H5T_COMPOUND && H5Tget_member_index(t, field) >= 0;
}

// one column of a table (compound dataset) as doubles; `width` = elements per row (array-valued columns: P1, jnu, coordinates)
std::vector<double> table_col(hid_t loc, const char *dset, const char *field, size_t *rows = nullptr, size_t *width = nullptr)
{
    if (!has_link(loc, dset)) throw Fail(fmt("table %s does not exist", dset));
    Hid d(H5Dopen2(loc, dset, H5P_DEFAULT), 2);
    Hid ft(H5Dget_type(d), 4);
    if (H5Tget_class(ft) != H5T_COMPOUND) throw Fail(fmt("%s is not a table", dset));
    int idx = H5Tget_member_index(ft, field);
    if (idx < 0) throw Fail(fmt("table %s has no column %s", dset, field));
    Hid mt(H5Tget_member_type(ft, (unsigned)idx), 4);
    size_t w = 1;
    hid_t elem = H5T_NATIVE_DOUBLE;
    Hid arr(-1, 4);
    if (H5Tget_class(mt) == H5T_ARRAY) {
        int nd = H5Tget_array_ndims(mt);
        std::vector<hsize_t> ad((size_t)nd);
        H5Tget_array_dims2(mt, ad.data());
        for (hsize_t x : ad) w *= (size_t)x;
        arr.id = H5Tarray_create2(H5T_NATIVE_DOUBLE, (unsigned)nd, ad.data());
        elem = arr.id;
    }
    Hid mem(H5Tcreate(H5T_COMPOUND, w * sizeof(double)), 4);
    H5Tinsert(mem, field, 0, elem);
    Hid s(H5Dget_space(d), 5);
    size_t n = (size_t)H5Sget_simple_extent_npoints(s);
    std::vector<double> v(n * w);
    if (n && H5Dread(d, mem, H5S_ALL, H5S_ALL, H5P_DEFAULT, v.data()) < 0) throw Fail(fmt("cannot read column %s of %s", field, dset));
    if (rows) *rows = n;
    if (width) *width = w;
    return v;
}

void put_attr_str(hid_t obj, const char *name, const std::string &v)
{
    Hid t(H5Tcopy(H5T_C_S1), 4);
    H5Tset_size(t, std::max<size_t>(v.size(), 1));
    H5Tset_strpad(t, H5T_STR_NULLPAD);
    Hid s(H5Screate(H5S_SCALAR), 5);
    Hid a(H5Acreate2(obj, name, t, s, H5P_DEFAULT, H5P_DEFAULT), 3);
    std::string buf = v;
    buf.resize(std::max<size_t>(v.size(), 1), '\0');
    if (a < 0 || H5Awrite(a, t, buf.data()) < 0) throw Fail(fmt("cannot write attribute %s", name));
}

void put_attr_dbl(hid_t obj, const char *name, double v)
{
    Hid s(H5Screate(H5S_SCALAR), 5);
    Hid a(H5Acreate2(obj, name, H5T_NATIVE_DOUBLE, s, H5P_DEFAULT, H5P_DEFAULT), 3);
    if (a < 0 || H5Awrite(a, H5T_NATIVE_DOUBLE, &v) < 0) throw Fail(fmt("cannot write attribute %s", name));
}

void put_attr_i32(hid_t obj, const char *name, int32_t v)
{
    Hid s(H5Screate(H5S_SCALAR), 5);
    Hid a(H5Acreate2(obj, name, H5T_NATIVE_INT32, s, H5P_DEFAULT, H5P_DEFAULT), 3);
    if (a < 0 || H5Awrite(a, H5T_NATIVE_INT32, &v) < 0) throw Fail(fmt("cannot write attribute %s", name));
}

// gzip-compressed dataset of doubles stored as float64 / float32 (physics_io_bytes, io_bytes) or int64; returns it open
hid_t put_dataset(hid_t loc, const char *name, const std::vector<hsize_t> &dims, const double *data, int kind /* 8, 4, or -8 = int64 */)
{
    Hid s(H5Screate_simple((int)dims.size(), dims.data(), nullptr), 5);
    Hid pl(H5Pcreate(H5P_DATASET_CREATE), 6);
    size_t n = 1;
    for (hsize_t x : dims) n *= (size_t)x;
    if (n > 0 && !dims.empty()) {
        // one chunk per slowest index keeps chunks below HDF5's 4 GiB limit for any grid this engine can hold
        std::vector<hsize_t> ch(dims);
        size_t bytes = n * 8;
        // (a few MB at most: a deflated chunk larger than HDF5's chunk cache is compressed and written in one piece)
        // leading dimensions collapse to 1 while the chunk is too large; the dimension that is left over is SPLIT, never
        // reduced to single elements: (n_dust, n_cells) of a large octree / Voronoi grid gives chunks of (1, 524288)
        const size_t limit = 4u << 20;
        for (size_t k = 0; k < ch.size() && bytes > limit; k++) {
            const size_t inner = bytes / (size_t)ch[k];       // bytes of one index of dimension k
            if (inner <= limit || k + 1 == ch.size()) {
                ch[k] = (hsize_t)std::max<size_t>(1, limit / std::max<size_t>(inner, 1));
                bytes = inner * (size_t)ch[k];
                break;
            }
            bytes = inner; ch[k] = 1;
        }
        H5Pset_chunk(pl, (int)ch.size(), ch.data());
        H5Pset_deflate(pl, 4);
    }
    const hid_t ftype = kind == 4 ? H5T_IEEE_F32LE : kind == -8 ? H5T_STD_I64LE : H5T_IEEE_F64LE;
    hid_t d = H5Dcreate2(loc, name, ftype, s, H5P_DEFAULT, pl, H5P_DEFAULT);
    if (d < 0) throw Fail(fmt("cannot create dataset %s", name));
    if (n && H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0) { H5Dclose(d); throw Fail(fmt("cannot write dataset %s", name)); }
    return d;
}

std::string now_string()        // "DD Month YYYY at HH:MM:SS"
{
    char buf[128];
    time_t t = time(nullptr);
    strftime(buf, sizeof buf, "%d %B %Y at %H:%M:%S", localtime(&t));
    return buf;
}

// ---------------------------------------------------------------------------------------------------------------
// The problem as read from the .rtin: owns every array the descriptors point to
// ---------------------------------------------------------------------------------------------------------------
struct Group {
    hyp_peeled_desc d;
    std::vector<double> theta, phi, filt_nu, filt_tr, filt_nu0;
    std::vector<int32_t> filt_n;
    std::string track_origin;
    double wav_min = 1.0, wav_max = 1000.0;
    int io_bytes = 8;
    bool binned = false;
};

struct Input {
    hyp_problem P;
    std::deque<std::vector<double>> keep;           // arrays referenced by the descriptors
    std::deque<std::vector<int32_t>> keep_i;
    std::deque<std::vector<hyp_spot_desc>> keep_spots;
    std::vector<hyp_dust_desc> dust;
    std::vector<hyp_source_desc> sources;
    std::vector<hyp_peeled_desc> groups_desc;       // peeled groups, then the binned one
    std::vector<Group> groups;
    std::string grid_type, geometry_id;
    std::vector<hsize_t> cell_dims;                 // shape of one species plane as written to the .rtout (n3, n2, n1) or (n_cells)
    std::vector<std::string> amr_paths;             // level_NNNNN/grid_NNNNN of every AMR grid
    // run control that is not part of the C ABI (src/main/setup_rt.f90)
    long long n_initial_iter = 0, n_initial_photons = 0, n_last_photons = 0, n_last_photons_sources = 0, n_last_photons_dust = 0,
              n_ray_photons_sources = 0, n_ray_photons_dust = 0;
    bool check_convergence = false, copy_input = false;
    double conv_abs = 0.0, conv_rel = 0.0, conv_pct = 100.0;
    std::string out_density = "none", out_density_diff = "none", out_specific_energy = "last", out_n_photons = "none", out_spectrum = "none";
    int physics_io_bytes = 8;
    std::vector<double> density;                    // kept for density_diff
    size_t n_cells = 0;
    const double *hold(std::vector<double> v) { keep.push_back(std::move(v)); return keep.back().data(); }
    const int32_t *hold_i(std::vector<int32_t> v) { keep_i.push_back(std::move(v)); return keep_i.back().data(); }
};


// version tuple of "0.9.10" / "0.8.7dev" (scripts and setup_rt.f90:38-45 refuse files older than 0.8.7)
std::vector<int> version_tuple(const std::string &v)
{
    std::vector<int> out;
    size_t i = 0;
    while (i < v.size()) {
        size_t j = v.find('.', i);
        std::string part = v.substr(i, j == std::string::npos ? std::string::npos : j - i);
        if (part.empty() || !isdigit((unsigned char)part[0])) break;
        std::string digits;
        for (char ch : part) if (isdigit((unsigned char)ch)) digits += ch;
        out.push_back(atoi(digits.c_str()));
        if (j == std::string::npos) break;
        i = j + 1;
    }
    return out;
}

void check_mode(const std::string &key, const std::string &v)
{
    if (v != "all" && v != "last" && v != "none") throw Fail(key + " should be one of all/last/none");
}

// /Dust/<name>: src/dust/dust_type_4elem.f90:78-293
void read_dust(Input &in, hid_t g, double minimum_specific_energy, hyp_dust_desc &x)
{
    std::memset(&x, 0, sizeof x);
    x.version = (int32_t)attr_int(g, "version");
    x.is_lte = attr_bool(g, "lte") ? 1 : 0;
    if (attr_str(g, "emissvar") != "E") throw Fail("Only emissvar='E' supported at this time");
    const std::string sub = attr_str(g, "sublimation_mode");
    x.sublimation_mode = sub == "no" ? 0 : sub == "fast" ? 1 : sub == "slow" ? 2 : sub == "cap" ? 3 : -1;
    if (x.sublimation_mode < 0) throw Fail("Unknown dust sublimation mode: " + sub);
    if (x.sublimation_mode != 0) x.sublimation_specific_energy = attr_dbl(g, "sublimation_specific_energy");
    x.minimum_specific_energy = minimum_specific_energy;
    size_t n = 0, w = 0;
    x.nu = in.hold(table_col(g, "optical_properties", "nu", &n)); x.n_nu = (int32_t)n;
    x.albedo = in.hold(table_col(g, "optical_properties", "albedo"));
    x.chi = in.hold(table_col(g, "optical_properties", "chi"));
    x.mu = in.hold(table_col(g, "scattering_angles", "mu", &n)); x.n_mu = (int32_t)n;
    x.P1 = in.hold(table_col(g, "optical_properties", "P1", nullptr, &w));
    if ((int32_t)w != x.n_mu) throw Fail("scattering matrix does not match the scattering angles");
    x.P2 = in.hold(table_col(g, "optical_properties", "P2"));
    x.P3 = in.hold(table_col(g, "optical_properties", "P3"));
    x.P4 = in.hold(table_col(g, "optical_properties", "P4"));
    x.emiss_nu = in.hold(table_col(g, "emissivities", "nu", &n)); x.n_enu = (int32_t)n;
    x.emiss_jnu = in.hold(table_col(g, "emissivities", "jnu", nullptr, &w)); x.n_jnu = (int32_t)w;
    x.emiss_var = in.hold(table_col(g, "emissivity_variable", "specific_energy", &n));
    if ((int32_t)n != x.n_jnu) throw Fail("emissivities do not match the emissivity variable");
    x.mo_specific_energy = in.hold(table_col(g, "mean_opacities", "specific_energy", &n)); x.n_e = (int32_t)n;
    x.mo_chi_rosseland = in.hold(table_col(g, "mean_opacities", "chi_rosseland"));
    if (table_has(g, "mean_opacities", "kappa_planck")) x.mo_kappa_planck = in.hold(table_col(g, "mean_opacities", "kappa_planck"));
    // dust_type_4elem.f90:231-237: version-1 files carry the Rosseland mean in the place of chi_inv_planck
    const char *inv = x.version == 1 ? "chi_rosseland" : "chi_inv_planck";
    if (table_has(g, "mean_opacities", inv)) x.mo_chi_inv_planck = in.hold(table_col(g, "mean_opacities", inv));
}

// one image group: image_setup (image_type.f90:153-335) + the peeled (images_peeled.f90:306-345) or binned
// (images_binned.f90:42-56) extras
void read_group(Input &in, hid_t g, bool binned, Group &G)
{
    hyp_peeled_desc &d = G.d;
    std::memset(&d, 0, sizeof d);
    G.binned = binned;
    const hyp_config &cfg = in.P.config;
    d.d_min = -std::numeric_limits<double>::infinity(); d.d_max = std::numeric_limits<double>::infinity();
    if (binned) {
        G.theta.assign(1, 0.0); G.phi.assign(1, 0.0);
    } else {
        G.theta = table_col(g, "angles", "theta"); G.phi = table_col(g, "angles", "phi");
        d.inside_observer = attr_bool(g, "inside_observer");
        d.ignore_optical_depth = attr_bool(g, "ignore_optical_depth");
        d.d_min = attr_dbl(g, "d_min"); d.d_max = attr_dbl(g, "d_max");
        const char *keys[2][3] = {{"peeloff_x", "peeloff_y", "peeloff_z"}, {"observer_x", "observer_y", "observer_z"}};
        for (int k = 0; k < 3; k++) d.peeloff_origin[k] = attr_dbl(g, keys[d.inside_observer ? 1 : 0][k]);
    }
    d.n_view = (int32_t)G.theta.size();
    const bool use_filters = has_attr(g, "use_filters") && attr_bool(g, "use_filters");
    if (use_filters) {      // image_type.f90:173-181,285-291
        if (cfg.monochromatic) throw Fail("cannot use filters in monochromatic mode");
        if (cfg.raytracing && !binned) throw Fail("filter convolution cannot be used with raytracing");     // images_peeled.f90:349-351
        const int nf = (int)attr_int(g, "n_filt");
        for (int i = 1; i <= nf; i++) {
            const std::string name = fmt("filter_%05d", i);
            size_t n = 0;
            std::vector<double> nu = table_col(g, name.c_str(), "nu", &n), tn = table_col(g, name.c_str(), "tn");
            G.filt_n.push_back((int32_t)n);
            G.filt_nu.insert(G.filt_nu.end(), nu.begin(), nu.end());
            G.filt_tr.insert(G.filt_tr.end(), tn.begin(), tn.end());
            Hid fd(H5Dopen2(g, name.c_str(), H5P_DEFAULT), 2);
            G.filt_nu0.push_back(attr_dbl(fd, "nu0"));
        }
        d.use_filters = 1; d.n_nu = nf;
    } else {
        d.n_nu = (int32_t)attr_int(g, "n_wav");
    }
    if (d.n_nu < 1) throw Fail("n_nu should be >= 1");
    G.io_bytes = has_attr(g, "io_bytes") ? (int)attr_int(g, "io_bytes") : 8;
    if (G.io_bytes != 4 && G.io_bytes != 8) throw Fail("unexpected value of io_bytes (should be 4 or 8)");
    if (use_filters) {
    } else if (cfg.monochromatic) {     // image_type.f90:243-258
        d.inu_min = (int32_t)attr_int(g, "inu_min"); d.inu_max = (int32_t)attr_int(g, "inu_max");
        d.n_nu = d.inu_max - d.inu_min + 1;
    } else {
        G.wav_min = attr_dbl(g, "wav_min"); G.wav_max = attr_dbl(g, "wav_max");
    }
    d.nu_min = C_CGS / (G.wav_max * 1.0e-4); d.nu_max = C_CGS / (G.wav_min * 1.0e-4);
    d.n_x = d.n_y = d.n_ap = 1;
    d.x_min = -1.0; d.x_max = 1.0; d.y_min = -1.0; d.y_max = 1.0; d.ap_min = d.ap_max = 1.0;
    d.compute_image = attr_bool(g, "compute_image");
    if (d.compute_image) {
        d.n_x = (int32_t)attr_int(g, "n_x"); d.n_y = (int32_t)attr_int(g, "n_y");
        d.x_min = attr_dbl(g, "x_min"); d.x_max = attr_dbl(g, "x_max"); d.y_min = attr_dbl(g, "y_min"); d.y_max = attr_dbl(g, "y_max");
    }
    d.compute_sed = attr_bool(g, "compute_sed");
    if (d.compute_sed) { d.n_ap = (int32_t)attr_int(g, "n_ap"); d.ap_min = attr_dbl(g, "ap_min"); d.ap_max = attr_dbl(g, "ap_max"); }
    G.track_origin = attr_str(g, "track_origin");
    d.track_origin = G.track_origin == "no" ? 0 : G.track_origin == "basic" ? 1 : G.track_origin == "detailed" ? 2 : G.track_origin == "scatterings" ? 3 : -1;
    if (d.track_origin < 0) throw Fail("unknown track_origin flag: " + G.track_origin);
    d.track_n_scat = has_attr(g, "track_n_scat") ? (int32_t)attr_int(g, "track_n_scat") : 0;
    d.uncertainties = attr_bool(g, "uncertainties");
    d.compute_stokes = has_attr(g, "compute_stokes") ? attr_bool(g, "compute_stokes") : 1;
}

// the whole .rtin
void read_rtin(const char *path, Input &in)
{
    Hid f(H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT), 0);
    if (f < 0) throw Fail(fmt("cannot open input file %s", path));
    hyp_problem &P = in.P;
    std::memset(&P, 0, sizeof P);
    hyp_config &c = P.config;
    Hid root(H5Gopen2(f, "/", H5P_DEFAULT), 1);
    // src/main/setup_rt.f90:38-45
    {
        const std::vector<int> v = has_attr(root, "python_version") ? version_tuple(attr_str(root, "python_version")) : std::vector<int>();
        const std::vector<int> need = {0, 8, 7};
        if (v.empty() || std::lexicographical_compare(v.begin(), v.end(), need.begin(), need.end()))
            throw Fail("cannot read files made with the Python module before version 0.8.7");
    }
    c.seed = has_attr(root, "seed") ? attr_int(root, "seed") : -124902;
    c.n_inter_max = attr_int(root, "n_inter_max");
    c.n_reabs_max = attr_int(root, "n_reabs_max");
    c.kill_on_absorb = attr_bool(root, "kill_on_absorb");
    c.kill_on_scatter = has_attr(root, "kill_on_scatter") ? attr_bool(root, "kill_on_scatter") : 0;
    c.sample_sources_evenly = has_attr(root, "sample_sources_evenly") ? attr_bool(root, "sample_sources_evenly") : 0;
    c.enforce_energy_range = has_attr(root, "enforce_energy_range") ? attr_bool(root, "enforce_energy_range") : 1;
    c.forced_first_interaction_algorithm = 1; c.baes16_xi = 0.5;
    if (has_attr(root, "forced_first_scattering")) c.forced_first_interaction = attr_bool(root, "forced_first_scattering");
    else {
        c.forced_first_interaction = attr_bool(root, "forced_first_interaction");
        const std::string algo = attr_str(root, "forced_first_interaction_algorithm");
        c.forced_first_interaction_algorithm = algo == "wr99" ? 1 : algo == "baes16" ? 2 : 0;
        if (!c.forced_first_interaction_algorithm) throw Fail("Unknown forced first interaction algorithm: " + algo);
        if (has_attr(root, "forced_first_interaction_baes16_xi")) c.baes16_xi = attr_dbl(root, "forced_first_interaction_baes16_xi");
    }
    c.propagation_check_frequency = has_attr(root, "propagation_check_frequency") ? attr_dbl(root, "propagation_check_frequency") : 1.0e-3;
    {
        const std::string t = has_attr(root, "specific_energy_type") ? attr_str(root, "specific_energy_type") : "initial";
        if (t != "initial" && t != "additional") throw Fail("specific_energy_type should be 'additional' or 'initial'");
        c.specific_energy_type = t == "additional";
    }
    in.n_initial_iter = attr_int(root, "n_initial_iter");
    in.n_initial_photons = in.n_initial_iter > 0 ? attr_int(root, "n_initial_photons") : 0;
    c.mrw = attr_bool(root, "mrw");
    c.mrw_gamma = 1.0; c.n_inter_mrw_max = 1000;
    if (c.mrw) { c.mrw_gamma = attr_dbl(root, "mrw_gamma"); c.n_inter_mrw_max = attr_int(root, "n_inter_mrw_max"); }
    c.pda = attr_bool(root, "pda");
    c.monochromatic = attr_bool(root, "monochromatic");
    c.raytracing = attr_bool(root, "raytracing");
    c.monochromatic_energy_threshold = 1.0e-10;
    if (c.monochromatic) {      // setup_rt.f90:49-57,220-222
        size_t n = 0;
        c.frequencies = in.hold(table_col(f, "frequencies", "nu", &n)); c.n_frequencies = (int32_t)n;
        if (has_attr(root, "monochromatic_energy_threshold")) c.monochromatic_energy_threshold = attr_dbl(root, "monochromatic_energy_threshold");
        in.n_last_photons_sources = has_attr(root, "n_last_photons_sources") ? attr_int(root, "n_last_photons_sources") : 0;
        in.n_last_photons_dust = has_attr(root, "n_last_photons_dust") ? attr_int(root, "n_last_photons_dust") : 0;
    }
    in.n_last_photons = has_attr(root, "n_last_photons") ? attr_int(root, "n_last_photons") : 0;
    if (c.raytracing) {
        in.n_ray_photons_sources = has_attr(root, "n_ray_photons_sources") ? attr_int(root, "n_ray_photons_sources") : 0;
        in.n_ray_photons_dust = has_attr(root, "n_ray_photons_dust") ? attr_int(root, "n_ray_photons_dust") : 0;
    }
    if (in.n_initial_iter > 0) {
        in.check_convergence = attr_bool(root, "check_convergence");
        if (in.check_convergence) {
            in.conv_abs = attr_dbl(root, "convergence_absolute"); in.conv_rel = attr_dbl(root, "convergence_relative");
            in.conv_pct = attr_dbl(root, "convergence_percentile");
        }
    }
    // setup_rt.f90:77-104,247-283: every /Output switch is one of all / last / none
    Hid out(H5Gopen2(f, "Output", H5P_DEFAULT), 1);
    if (out < 0) throw Fail("group Output does not exist");
    in.out_density = attr_str(out, "output_density"); check_mode("output_density", in.out_density);
    in.out_density_diff = attr_str(out, "output_density_diff"); check_mode("output_density_diff", in.out_density_diff);
    in.out_specific_energy = attr_str(out, "output_specific_energy"); check_mode("output_specific_energy", in.out_specific_energy);
    in.out_n_photons = attr_str(out, "output_n_photons"); check_mode("output_n_photons", in.out_n_photons);
    if (has_attr(out, "output_specific_energy_spectrum")) {
        in.out_spectrum = attr_str(out, "output_specific_energy_spectrum");
        check_mode("output_specific_energy_spectrum", in.out_spectrum);
    }
    c.count_photons = c.pda || in.out_n_photons != "none";
    if (in.out_spectrum != "none") {
        if (!has_link(f, "specific_energy_spectrum_bin_edges"))
            throw Fail("specific_energy_spectrum_bin_edges should be present in the input when output_specific_energy_spectrum is enabled");
        size_t n = 0;
        std::vector<double> e = table_col(f, "specific_energy_spectrum_bin_edges", "nu", &n);
        for (size_t i = 1; i < n; i++) if (!(e[i] > e[i - 1])) throw Fail("specific_energy_spectrum_bin_edges should be strictly increasing");
        if (n < 2) throw Fail("specific_energy_spectrum_bin_edges should be strictly increasing");
        c.n_spectrum_bins = (int32_t)n - 1;
        c.spectrum_bin_edges = in.hold(std::move(e));
    }
    // setup_rt.f90:207-215, main.f90:133-150
    in.physics_io_bytes = has_attr(root, "physics_io_bytes") ? (int)attr_int(root, "physics_io_bytes") : 8;
    if (in.physics_io_bytes != 4 && in.physics_io_bytes != 8) throw Fail("unexpected value of physics_io_bytes (should be 4 or 8)");
    in.copy_input = has_attr(root, "copy_input") ? attr_bool(root, "copy_input") : false;

    // ---- geometry ----
    Hid geo(H5Gopen2(f, "Grid/Geometry", H5P_DEFAULT), 1);
    if (geo < 0) throw Fail("group Grid/Geometry does not exist");
    hyp_grid_desc &G = P.grid;
    in.grid_type = attr_str(geo, "grid_type");
    in.geometry_id = has_attr(geo, "geometry") ? attr_str(geo, "geometry") : "";
    const std::string &gt = in.grid_type;
    if (gt == "car" || gt == "sph_pol" || gt == "cyl_pol") {
        const char *cols[3] = {gt == "car" ? "x" : gt == "sph_pol" ? "r" : "w", gt == "car" ? "y" : gt == "sph_pol" ? "t" : "z", gt == "car" ? "z" : "p"};
        size_t n1 = 0, n2 = 0, n3 = 0;
        G.w1 = in.hold(table_col(geo, "walls_1", cols[0], &n1));
        G.w2 = in.hold(table_col(geo, "walls_2", cols[1], &n2));
        G.w3 = in.hold(table_col(geo, "walls_3", cols[2], &n3));
        if (n1 < 2 || n2 < 2 || n3 < 2) throw Fail("grid needs at least one cell per axis");
        G.type = gt == "car" ? 1 : gt == "sph_pol" ? 5 : 6;
        G.n1 = (int32_t)n1 - 1; G.n2 = (int32_t)n2 - 1; G.n3 = (int32_t)n3 - 1;
        in.n_cells = (size_t)G.n1 * G.n2 * G.n3;
        G.n_cells = (int64_t)in.n_cells;
        in.cell_dims = {(hsize_t)G.n3, (hsize_t)G.n2, (hsize_t)G.n1};
    } else if (gt == "oct") {
        G.type = 2;
        std::vector<double> r = table_col(geo, "cells", "refined");
        std::vector<int32_t> ref(r.size());
        for (size_t i = 0; i < r.size(); i++) ref[i] = (int32_t)r[i];
        if (ref.empty() || (ref.size() - 1) % 8 != 0) throw Fail("refined should have shape 8 * n + 1");
        in.n_cells = ref.size(); G.n_cells = (int64_t)ref.size();
        G.refined = in.hold_i(std::move(ref));
        const char *ck[3] = {"x", "y", "z"}, *hk[3] = {"dx", "dy", "dz"};
        for (int k = 0; k < 3; k++) { G.oct_center[k] = attr_dbl(geo, ck[k]); G.oct_half[k] = attr_dbl(geo, hk[k]); }
        in.cell_dims = {(hsize_t)in.n_cells};
    } else if (gt == "vor") {
        G.type = 3;
        size_t n = 0, w = 0;
        G.vor_sites = in.hold(table_col(geo, "cells", "coordinates", &n, &w));
        if (w != 3) throw Fail("Voronoi sites should have three coordinates");
        G.vor_volume = in.hold(table_col(geo, "cells", "volume"));
        if (table_has(geo, "cells", "bb_min")) {
            std::vector<double> lo = table_col(geo, "cells", "bb_min"), hi = table_col(geo, "cells", "bb_max"), bb(6 * n);
            for (size_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) { bb[6 * i + k] = lo[3 * i + k]; bb[6 * i + 3 + k] = hi[3 * i + k]; }
            G.vor_bb = in.hold(std::move(bb));
        }
        std::vector<int32_t> idx = read_ints(geo, "sparse_idx"), nei = read_ints(geo, "sparse_neighs");
        if (idx.size() != n + 1 || (size_t)idx.back() != nei.size()) throw Fail("inconsistent Voronoi neighbour lists");
        G.vor_idx = in.hold_i(std::move(idx)); G.vor_neighs = in.hold_i(std::move(nei));
        const char *bk[6] = {"xmin", "xmax", "ymin", "ymax", "zmin", "zmax"};
        for (int k = 0; k < 6; k++) G.vor_box[k] = attr_dbl(geo, bk[k]);
        in.n_cells = n; G.n_cells = (int64_t)n;
        in.cell_dims = {(hsize_t)n};
    } else if (gt == "amr") {
        G.type = 4;
        std::vector<int32_t> lev, nn;
        std::vector<double> bounds;
        const int nlev = (int)attr_int(geo, "nlevels");
        size_t total = 0;
        for (int il = 1; il <= nlev; il++) {
            const std::string ln = fmt("level_%05d", il);
            Hid gl(H5Gopen2(geo, ln.c_str(), H5P_DEFAULT), 1);
            if (gl < 0) throw Fail("missing AMR level group " + ln);
            const int ng = (int)attr_int(gl, "ngrids");
            for (int ig = 1; ig <= ng; ig++) {
                const std::string gn = fmt("grid_%05d", ig);
                Hid gg(H5Gopen2(gl, gn.c_str(), H5P_DEFAULT), 1);
                if (gg < 0) throw Fail("missing AMR grid group " + ln + "/" + gn);
                lev.push_back(il);
                const int a1 = (int)attr_int(gg, "n1"), a2 = (int)attr_int(gg, "n2"), a3 = (int)attr_int(gg, "n3");
                nn.push_back(a1); nn.push_back(a2); nn.push_back(a3);
                const char *bk[6] = {"xmin", "xmax", "ymin", "ymax", "zmin", "zmax"};
                for (int k = 0; k < 6; k++) bounds.push_back(attr_dbl(gg, bk[k]));
                in.amr_paths.push_back(ln + "/" + gn);
                total += (size_t)a1 * a2 * a3;
            }
        }
        G.n_amr_levels = nlev; G.n_amr_grids = (int32_t)lev.size();
        G.amr_level = in.hold_i(std::move(lev)); G.amr_n = in.hold_i(std::move(nn)); G.amr_bounds = in.hold(std::move(bounds));
        in.n_cells = total; G.n_cells = (int64_t)total;
        in.cell_dims = {(hsize_t)total};
    } else throw Fail("Unexpected coordinate type: " + gt);

    // ---- density, specific energy ----
    Hid q(H5Gopen2(f, "Grid/Quantities", H5P_DEFAULT), 1);
    if (q < 0) throw Fail("group Grid/Quantities does not exist");
    size_t n_dust = 0;
    std::vector<double> spec;
    if (gt == "amr") {
        // read_grid_4d for AMR (src/grid/grid_io_amr_template.f90): one (n_dust, n3, n2, n1) array per grid
        auto gather = [&](const char *name, std::vector<double> &dst) {
            std::vector<std::vector<double>> parts;
            std::vector<size_t> sizes;
            for (const std::string &pth : in.amr_paths) {
                Hid gg(H5Gopen2(q, pth.c_str(), H5P_DEFAULT), 1);
                if (gg < 0) throw Fail("missing quantities of AMR grid " + pth);
                std::vector<hsize_t> dm;
                parts.push_back(read_doubles(gg, name, &dm));
                if (dm.size() != 4) throw Fail(std::string(name) + " of an AMR grid should have four dimensions");
                if (n_dust == 0) n_dust = (size_t)dm[0];
                if ((size_t)dm[0] != n_dust) throw Fail("density array has wrong number of dust types");
                sizes.push_back(parts.back().size() / n_dust);
            }
            dst.assign(n_dust * in.n_cells, 0.0);
            size_t off = 0;
            for (size_t k = 0; k < parts.size(); k++) {
                for (size_t d = 0; d < n_dust; d++)
                    std::copy(parts[k].begin() + (long)(d * sizes[k]), parts[k].begin() + (long)((d + 1) * sizes[k]), dst.begin() + (long)(d * in.n_cells + off));
                off += sizes[k];
            }
            if (off != in.n_cells) throw Fail("AMR quantities do not match the geometry");
        };
        gather("density", in.density);
        Hid g0(H5Gopen2(q, in.amr_paths[0].c_str(), H5P_DEFAULT), 1);
        if (has_link(g0, "specific_energy")) gather("specific_energy", spec);
    } else {
        std::vector<hsize_t> dm;
        in.density = read_doubles(q, "density", &dm);
        if (dm.empty()) throw Fail("density array has wrong shape");
        n_dust = (size_t)dm[0];
        if (in.density.size() != n_dust * in.n_cells) throw Fail("density array has wrong shape");
        if (has_link(q, "specific_energy")) {
            spec = read_doubles(q, "specific_energy");
            if (spec.size() != in.density.size()) throw Fail("specific_energy array has wrong number of dust types");
        }
        // read_grid_4d (src/grid/grid_io.f90:78-81): the dataset must belong to this geometry
        for (const char *name : {"density", "specific_energy"}) {
            if (!has_link(q, name)) continue;
            Hid d(H5Dopen2(q, name, H5P_DEFAULT), 2);
            if (has_attr(d, "geometry") && has_attr(geo, "geometry") && attr_str(d, "geometry") != in.geometry_id)
                throw Fail(std::string("geometry IDs do not match for ") + name);
        }
    }
    if (n_dust > HYP_MAX_DUST) throw Fail(fmt("at most %d dust species are supported", HYP_MAX_DUST));
    std::vector<double> mse(n_dust, 0.0);
    if (has_attr(q, "minimum_specific_energy")) {
        std::vector<double> m = attr_dbl_array(q, "minimum_specific_energy");
        for (size_t i = 0; i < n_dust && i < m.size(); i++) mse[i] = m[i];
    }
    P.density = in.density.data();
    if (!spec.empty()) P.specific_energy = in.hold(std::move(spec));

    // ---- dust ----
    Hid dg(H5Gopen2(f, "Dust", H5P_DEFAULT), 1);
    std::vector<std::string> dnames = dg < 0 ? std::vector<std::string>() : children(dg);
    if (dnames.size() != n_dust) throw Fail("density array has wrong number of dust types");
    in.dust.resize(n_dust);
    for (size_t i = 0; i < n_dust; i++) {
        Hid g(H5Gopen2(dg, dnames[i].c_str(), H5P_DEFAULT), 1);
        read_dust(in, g, mse[i], in.dust[i]);
    }
    P.n_dust = (int32_t)n_dust; P.dust = in.dust.data();

    // ---- sources: src/sources/source_type.f90:102-322 ----
    Hid sg(H5Gopen2(f, "Sources", H5P_DEFAULT), 1);
    std::vector<std::string> snames = sg < 0 ? std::vector<std::string>() : children(sg);
    in.sources.resize(snames.size());
    for (size_t i = 0; i < snames.size(); i++) {
        Hid g(H5Gopen2(sg, snames[i].c_str(), H5P_DEFAULT), 1);
        hyp_source_desc &x = in.sources[i];
        std::memset(&x, 0, sizeof x);
        const std::string t = attr_str(g, "type");
        x.peeloff = attr_bool(g, "peeloff");
        if (t == "point_collection") {
            x.type = 8;
            size_t n = 0;
            std::vector<double> lum = read_doubles(g, "luminosity");
            n = lum.size();
            double s = 0.0;
            for (double v : lum) s += v;
            x.luminosity = s; x.n_points = (int32_t)n;
            x.point_lum = in.hold(std::move(lum));
            std::vector<double> pos = read_doubles(g, "position");
            if (pos.size() != 3 * n) throw Fail("point collection needs one position per luminosity");
            x.points = in.hold(std::move(pos));
        } else x.luminosity = attr_dbl(g, "luminosity");
        auto position = [&]() { x.position[0] = attr_dbl(g, "x"); x.position[1] = attr_dbl(g, "y"); x.position[2] = attr_dbl(g, "z"); };
        if (t == "point") { x.type = 1; position(); }
        else if (t == "sphere") {
            x.type = 2; position(); x.radius = attr_dbl(g, "r"); x.limb_darkening = attr_bool(g, "limb");
            // spots are sub-groups of the source group (source_type.f90:150-188)
            std::vector<hyp_spot_desc> spots;
            for (const std::string &k : children(g)) {
                if (!is_group(g, k)) continue;
                Hid sp(H5Gopen2(g, k.c_str(), H5P_DEFAULT), 1);
                hyp_spot_desc q1;
                std::memset(&q1, 0, sizeof q1);
                q1.longitude = attr_dbl(sp, "longitude"); q1.latitude = attr_dbl(sp, "latitude"); q1.radius = attr_dbl(sp, "radius");
                q1.luminosity = attr_dbl(sp, "luminosity");
                const std::string qs = attr_str(sp, "spectrum");
                if (qs == "temperature") { q1.spectrum_type = 2; q1.temperature = attr_dbl(sp, "temperature"); }
                else if (qs == "spectrum") {
                    size_t n = 0;
                    q1.spectrum_type = 1; q1.spec_nu = in.hold(table_col(sp, "spectrum", "nu", &n)); q1.spec_fnu = in.hold(table_col(sp, "spectrum", "fnu"));
                    q1.n_spec = (int32_t)n;
                } else throw Fail("Spot cannot have LTE spectrum");
                spots.push_back(q1);
            }
            if (!spots.empty()) { in.keep_spots.push_back(std::move(spots)); x.n_spots = (int32_t)in.keep_spots.back().size(); x.spots = in.keep_spots.back().data(); }
        }
        else if (t == "extern_sph") { x.type = 5; position(); x.radius = attr_dbl(g, "r"); }
        else if (t == "plane_parallel") { x.type = 7; position(); x.radius = attr_dbl(g, "r"); x.direction[0] = attr_dbl(g, "theta"); x.direction[1] = attr_dbl(g, "phi"); }
        else if (t == "point_collection") {}
        else if (t == "map") {      // source_type.f90:190-199; one dataset per AMR grid otherwise (grid_io_amr.f90)
            x.type = 4;
            std::vector<double> m;
            if (gt == "amr") {
                for (const std::string &pth : in.amr_paths) {
                    Hid gg(H5Gopen2(g, pth.c_str(), H5P_DEFAULT), 1);
                    if (gg < 0) throw Fail("missing luminosity map of AMR grid " + pth);
                    std::vector<double> part = read_doubles(gg, "Luminosity map");
                    m.insert(m.end(), part.begin(), part.end());
                }
            } else m = read_doubles(g, "Luminosity map");
            if (m.size() != in.n_cells) throw Fail("map source needs a luminosity map with one value per cell");
            x.map = in.hold(std::move(m));
        }
        else if (t == "extern_box") {
            x.type = 6;
            const char *bk[6] = {"xmin", "xmax", "ymin", "ymax", "zmin", "zmax"};
            for (int k = 0; k < 6; k++) x.box[k] = attr_dbl(g, bk[k]);
        }
        else throw Fail("unknown type in source list: " + t);
        const std::string st = attr_str(g, "spectrum");
        if (st == "temperature") { x.spectrum_type = 2; x.temperature = attr_dbl(g, "temperature"); }
        else if (st == "spectrum") {
            size_t n = 0;
            x.spectrum_type = 1; x.spec_nu = in.hold(table_col(g, "spectrum", "nu", &n)); x.spec_fnu = in.hold(table_col(g, "spectrum", "fnu"));
            x.n_spec = (int32_t)n;
        } else if (st == "lte" && t == "map") x.spectrum_type = 3;
        else throw Fail("Point source cannot have LTE spectrum");
    }
    P.n_sources = (int32_t)in.sources.size(); P.sources = in.sources.data();

    // ---- image groups ----
    if (has_link(out, "Peeled")) {
        Hid pg(H5Gopen2(out, "Peeled", H5P_DEFAULT), 1);
        for (const std::string &n : children(pg)) {
            Hid g(H5Gopen2(pg, n.c_str(), H5P_DEFAULT), 1);
            in.groups.emplace_back();
            read_group(in, g, false, in.groups.back());
        }
    }
    P.n_peeled = (int32_t)in.groups.size();
    if (has_link(out, "Binned")) {
        Hid bg(H5Gopen2(out, "Binned", H5P_DEFAULT), 1);
        std::vector<std::string> names = children(bg);
        if (names.size() > 1) throw Fail("can't have more than one binned image group");      // setup_rt.f90:324
        if (names.size() == 1) {
            Hid g(H5Gopen2(bg, names[0].c_str(), H5P_DEFAULT), 1);
            in.groups.emplace_back();
            read_group(in, g, true, in.groups.back());
            P.n_binned_theta = (int32_t)attr_int(g, "n_theta"); P.n_binned_phi = (int32_t)attr_int(g, "n_phi");
        }
    }
    for (Group &G2 : in.groups) {       // the vectors do not move any more
        G2.d.theta = G2.theta.data(); G2.d.phi = G2.phi.data();
        if (G2.d.use_filters) { G2.d.filt_n = G2.filt_n.data(); G2.d.filt_nu = G2.filt_nu.data(); G2.d.filt_tr = G2.filt_tr.data(); }
        if (c.monochromatic && G2.d.inu_min == 0 && G2.d.inu_max == 0) { G2.d.inu_min = 1; G2.d.inu_max = c.n_frequencies; G2.d.n_nu = c.n_frequencies; }
        in.groups_desc.push_back(G2.d);
    }
    P.peeled = in.groups_desc.data();
    if ((size_t)P.n_peeled < in.groups.size()) P.binned = &in.groups_desc[(size_t)P.n_peeled];
}

}  // namespace

namespace {

bool want(const std::string &mode, long long it, long long last) { return mode == "all" || (mode == "last" && it == last); }

void check(int rc, hyp_handle h)
{
    if (rc != 0) { const char *m = hyp_last_error(h); throw Fail(m && *m ? m : "engine error"); }
}


// ---------------------------------------------------------------------------------------------------------------
// Ranks: one process per GPU, RCCL for the one collective of each iteration (src/mpi/mpi_routines.f90:272-361)
// ---------------------------------------------------------------------------------------------------------------
struct Comm {
    int rank = 0, size = 1, local_rank = -1;
    bool on = false;                 // a launcher named a rank (also at size 1: the collective is then executed all the same)
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    double *zero = nullptr; size_t zero_n = 0;
    int ranks_seen = 0;              // result of the first collective: a sum of ones
    std::string id_file, abort_file;
    // What the watcher thread reads: its own copies, shared with this object, so that the thread (detached: it may be asleep when
    // main returns) never touches a Comm that is gone.  `leaving`: this rank wrote the abort file itself, or finished -- its
    // watcher stands down.
    struct Watch { std::string id_file, abort_file; int rank = 0; std::atomic<bool> leaving{false}; };
    std::shared_ptr<Watch> w = std::make_shared<Watch>();

    static bool env_int(std::initializer_list<const char *> names, int &out)
    {
        for (const char *n : names) { const char *v = getenv(n); if (v && *v) { out = atoi(v); return true; } }
        return false;
    }
    // `mpi_name`: started under one of the reference's hyperion_<grid>_mpi names (scripts/hyperion:62-92).  An explicit RANK /
    // WORLD_SIZE (torchrun, --ranks) always counts; the variables a batch system sets for ANY process of a job (Open MPI, PMI,
    // Slurm) make this process a rank only under the _mpi names -- `hyperion_car` inside an sbatch script with --ntasks=N is one
    // plain process, not rank 0 of N waiting for peers that do not exist.
    void from_env(const std::string &output, bool mpi_name)
    {
        on = env_int({"RANK"}, rank);
        if (on) { if (!env_int({"WORLD_SIZE"}, size)) size = 1; }
        else if (mpi_name) {
            on = env_int({"OMPI_COMM_WORLD_RANK", "PMI_RANK", "SLURM_PROCID"}, rank);
            if (on && !env_int({"OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "SLURM_NTASKS"}, size)) size = 1;
        }
        env_int({"HYP_DEVICE", "LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID"}, local_rank);
        if (!on) { rank = 0; size = 1; }
        if (rank < 0 || rank >= size) throw Fail(fmt("rank %d of %d: inconsistent launcher environment", rank, size));
        // the files the ranks of ONE launch talk through before (and beside) the communicator carry the launcher's id in their
        // names: a file a crashed run left behind belongs to another launch and is never read
        const char *e = getenv("HYP_NCCL_ID_FILE");
        id_file = (e && *e ? std::string(e) : output + ".ncclid") + "." + std::to_string(launch_id());
        abort_file = output + ".abort." + std::to_string(launch_id());
        w->id_file = id_file; w->abort_file = abort_file; w->rank = rank;
        if (on && size > 1) { std::shared_ptr<Watch> ws = w; std::thread([ws] { watch(ws); }).detach(); }
    }
    bool main_process() const { return rank == 0; }

    // The ncclUniqueId travels through a file next to the output (the ranks of one node share it): rank 0 writes
    // {magic, launch id, id} to a temporary name and renames it; the others wait for it.
    struct IdFile { char magic[8]; long long ppid; int aborted; char msg[256]; ncclUniqueId id; };
    // what the ranks of one launch have in common: the launcher's pid (--ranks sets HYP_LAUNCHER_PID; mpirun / torchrun: the parent)
    static long long launch_id() { const char *v = getenv("HYP_LAUNCHER_PID"); return v && *v ? atoll(v) : (long long)getppid(); }
    void publish(const IdFile &f) const
    {
        const std::string tmp = id_file + ".tmp";
        FILE *fp = fopen(tmp.c_str(), "wb");
        if (!fp || fwrite(&f, sizeof f, 1, fp) != 1) { if (fp) fclose(fp); throw Fail("cannot write " + tmp); }
        fclose(fp);
        if (rename(tmp.c_str(), id_file.c_str()) != 0) throw Fail("cannot publish " + id_file);
    }
    // error() / mp_stop of the reference (src/mpi/mpi_core.f90): a rank in error takes every rank down.  A rank that fails
    // OUTSIDE a collective it can still take part in (before the communicator exists, while reading the input, in hyp_create)
    // leaves `abort_file` with its message; every rank runs a watcher thread that finds it within a tenth of a second, repeats
    // the message and leaves with status 1 -- also out of ncclCommInitRank or an all-reduce that would never return.  The rank
    // that wrote the file removes it again after a grace period (and the unique-id file with it).
    void abort_others(const std::string &why)
    {
        if (!on || size == 1 || abort_file.empty()) return;
        w->leaving = true;       // BEFORE the file appears: this rank's own watcher must not report it as another rank's
        const std::string tmp = abort_file + fmt(".%d.tmp", rank), mine = fmt("rank %d: ", rank);
        FILE *fp = fopen(tmp.c_str(), "w");
        if (fp) { fprintf(fp, "%s%s\n", mine.c_str(), why.c_str()); fclose(fp); (void)rename(tmp.c_str(), abort_file.c_str()); }
        usleep(1500000);
        // several ranks may fail together and each renames its own message over the file: only the writer of what the file holds
        // NOW removes it, so that a later rank's message stays for its full grace period
        char head[64] = "";
        fp = fopen(abort_file.c_str(), "r");
        if (fp) { if (!fgets(head, sizeof head, fp)) head[0] = 0; fclose(fp); }
        if (!std::strncmp(head, mine.c_str(), mine.size())) unlink(abort_file.c_str());
        if (rank == 0) unlink(id_file.c_str());
    }
    static void watch(std::shared_ptr<Watch> ws)
    {
        for (;;) {
            usleep(100000);
            if (ws->leaving) return;
            FILE *fp = fopen(ws->abort_file.c_str(), "r");
            if (!fp) continue;
            char msg[512] = "";
            if (!fgets(msg, sizeof msg, fp)) msg[0] = 0;
            fclose(fp);
            if (ws->leaving) return;
            fprintf(stderr, " ERROR: another rank stopped the run: %sAn error occurred, and the run did not complete\n", msg);
            if (ws->rank == 0) unlink(ws->id_file.c_str());
            _exit(1);
        }
    }
    void init(const std::string &output, int device)
    {
        if (!on) return;
        (void)output;
        if (hipSetDevice(device) != hipSuccess) throw Fail(fmt("hipSetDevice(%d) failed", device));
        IdFile f; std::memset(&f, 0, sizeof f);
        if (rank == 0) {
            std::memcpy(f.magic, "HYPNCCL", 8); f.ppid = launch_id();
            if (ncclGetUniqueId(&f.id) != ncclSuccess) throw Fail("ncclGetUniqueId failed");
            if (size > 1) publish(f);
        } else {
            const double t_end = (double)time(nullptr) + 300.0;
            for (;;) {
                FILE *fp = fopen(id_file.c_str(), "rb");
                bool ok = false;
                if (fp) { ok = fread(&f, sizeof f, 1, fp) == 1 && !std::memcmp(f.magic, "HYPNCCL", 8) && f.ppid == launch_id(); fclose(fp); }
                if (ok) break;
                if ((double)time(nullptr) > t_end) throw Fail("rank 0 did not publish " + id_file + " (ranks must share a launcher and a file system)");
                usleep(20000);
            }
        }
        if (ncclCommInitRank(&comm, size, f.id, rank) != ncclSuccess) throw Fail("ncclCommInitRank failed");
        if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) throw Fail("cannot create the collective's stream");
        // everybody has read the id once this first collective returns: rank 0 removes the file
        double *one = nullptr;
        if (hipMalloc((void **)&one, sizeof(double)) != hipSuccess) throw Fail("hipMalloc failed");
        // ... and it is a sum of ones: what comes back is the number of ranks RCCL actually joined (mp_initialize prints the
        // same of MPI_Comm_size, src/mpi/mpi_core.f90); anything but `size` is an error, not a slow run on fewer GPUs
        const double unit = 1.0;
        (void)hipMemcpy(one, &unit, sizeof unit, hipMemcpyHostToDevice);
        sum(one, 1);
        double seen = 0.0;
        (void)hipMemcpy(&seen, one, sizeof seen, hipMemcpyDeviceToHost);
        (void)hipFree(one);
        ranks_seen = (int)(seen + 0.5);
        if (ranks_seen != size) throw Fail(fmt("the all-reduce joined %d ranks, the launcher named %d", ranks_seen, size));
        char bus[32] = "";
        (void)hipDeviceGetPCIBusId(bus, sizeof bus, device);
        if (size > 1 || getenv("HYP_VERBOSE_RANKS")) printf(" [mpi] rank %d of %d: device %d (%s); all-reduce of ones over RCCL = %d ranks seen\n", rank, size, device, bus, ranks_seen);
        if (rank == 0 && size > 1) unlink(id_file.c_str());
    }
    void sum(void *ptr, size_t n)
    {
        if (ncclAllReduce(ptr, ptr, n, ncclDouble, ncclSum, comm, stream) != ncclSuccess) throw Fail("ncclAllReduce failed");
        if (hipStreamSynchronize(stream) != hipSuccess) throw Fail("the all-reduce failed");
    }
    // [first, first + n) of this rank: static id ranges (hyperion_amd/distributed.py: shard_range)
    void range(uint64_t n_total, uint64_t &first, uint64_t &n) const
    {
        first = n_total * (uint64_t)rank / (uint64_t)size;
        n = n_total * (uint64_t)(rank + 1) / (uint64_t)size - first;
    }
    // Sum the block of the iteration just launched over the ranks.  rc / err: what the launches of this rank returned.  A rank
    // in error contributes zeros with 1 in the TAIL_RANK_ERROR slot of the block's tail (`flag_opt` / `len_opt` name the
    // engine options that hold its index and the block's length), raises after the sum; the others are told by `finish`.
    void collect(hyp_handle h, int rc, int (*accumulators)(hyp_handle, void **, uint64_t *), const char *flag_opt, const char *len_opt)
    {
        std::string err;
        if (rc) { const char *m = hyp_last_error(h); err = m && *m ? m : "engine error"; }
        void *ptr = nullptr; uint64_t n = 0;
        if (!rc && accumulators(h, &ptr, &n) != 0) { const char *m = hyp_last_error(h); err = m && *m ? m : "engine error"; rc = 1; }
        if (!on) { if (rc) throw Fail(err); return; }
        if (rc) {
            int64_t len = 0, flag = 0;
            if (hyp_get_option(h, len_opt, &len) || hyp_get_option(h, flag_opt, &flag)) throw Fail(err);
            if (zero_n < (size_t)len) { if (zero) (void)hipFree(zero); zero = nullptr; if (hipMalloc((void **)&zero, sizeof(double) * (size_t)len) != hipSuccess) throw Fail(err); zero_n = (size_t)len; }
            (void)hipMemset(zero, 0, sizeof(double) * (size_t)len);
            const double one = 1.0;
            (void)hipMemcpy(zero + flag, &one, sizeof one, hipMemcpyHostToDevice);
            ptr = zero; n = (uint64_t)len;
        }
        sum(ptr, (size_t)n);
        if (rc) throw Fail(err);
    }
    void finalize()
    {
        w->leaving = true;
        if (zero) (void)hipFree(zero);
        if (comm) ncclCommDestroy(comm);
        if (stream) (void)hipStreamDestroy(stream);
        comm = nullptr; stream = nullptr; zero = nullptr;
    }
};

// the iterations through their split entry points (launch on this rank's id range / one all-reduce / finish everywhere)
void lucy_iteration(hyp_handle h, Comm &c, uint64_t n_total, int it, double *se_out, hyp_iter_stats *st)
{
    if (!c.on) { check(hyp_lucy_iteration(h, n_total, it, se_out, st), h); return; }
    if (n_total == 0) { *st = hyp_iter_stats{}; return; }
    uint64_t first, n; c.range(n_total, first, n);
    c.collect(h, hyp_lucy_launch(h, first, n, it), hyp_lucy_accumulators, "lucy_flag_index", "lucy_block_doubles");
    check(hyp_lucy_finish(h, se_out, st), h);
    st->n_packets = n_total;
}

void final_iteration(hyp_handle h, Comm &c, uint64_t n_total, hyp_iter_stats *st)
{
    if (!c.on) { check(hyp_final_iteration(h, n_total, st), h); return; }
    uint64_t first, n; c.range(n_total, first, n);
    c.collect(h, hyp_final_launch(h, first, n), hyp_final_accumulators, "image_flag_index", "image_block_doubles");
    check(hyp_final_finish(h, st), h);
    st->n_packets = n_total;
}

void raytracing_iteration(hyp_handle h, Comm &c, uint64_t n_src, uint64_t n_dust, hyp_iter_stats *st)
{
    if (!c.on) { check(hyp_raytracing_iteration(h, n_src, n_dust, st), h); return; }
    // rank 0 keeps the cubes of the final iteration, the others start from zero: one sum gives final + raytraced flux
    int rc = 0;
    const uint64_t tot[2] = {n_src, n_dust};
    for (int which = 0; which < 2 && !rc; which++) {
        uint64_t first, n; c.range(tot[which], first, n);
        rc = hyp_raytracing_launch(h, which, first, n, tot[which], which == 0 && c.rank > 0);
    }
    c.collect(h, rc, hyp_raytracing_accumulators, "image_flag_index", "image_block_doubles");
    check(hyp_raytracing_finish(h, st), h);
}

void mono_iteration(hyp_handle h, Comm &c, uint64_t n_src, uint64_t n_dust, int n_freq, hyp_iter_stats *st)
{
    if (!c.on) { check(hyp_mono_iteration(h, n_src, n_dust, st), h); return; }
    int rc = 0; bool first_launch = true;
    const uint64_t tot[2] = {n_src, n_dust};
    for (int which = 0; which < 2; which++) {
        uint64_t first, n; c.range(tot[which], first, n);
        for (int inu = 0; inu < n_freq; inu++) {
            if (!rc) rc = hyp_mono_launch(h, which, inu, first, n, tot[which], first_launch);
            first_launch = false;
        }
    }
    c.collect(h, rc, hyp_mono_accumulators, "image_flag_index", "image_block_doubles");
    check(hyp_mono_finish(h, st), h);
}

struct Iteration {
    long long index = 0;
    uint64_t killed_geo = 0, killed_int = 0;
    std::vector<double> specific_energy, density, density_diff, n_photons, spectrum, spectrum_edges;
};

// image_write (src/images/image_type.f90:608-788): raw flux sums -> nu F_nu (divide by the relative bin width, or multiply
// by nu at exact frequencies), uncertainties sqrt(sum x^2), SED apertures accumulated outwards
struct Cubes {
    std::vector<double> seds, seds_unc, images, images_unc;
    std::vector<hsize_t> sed_dims, img_dims;
};

Cubes fetch_group(hyp_handle h, const Input &in, size_t ig, const std::vector<double> *frequencies)
{
    const Group &G = in.groups[ig];
    const hyp_peeled_desc &d = in.groups_desc[ig];
    const int n_orig = hyp_peeled_n_orig(h, (int)ig);
    if (n_orig < 1) throw Fail("cannot query the image group");
    const size_t ns = d.compute_stokes ? 4 : 1;
    const size_t n_view = G.binned ? (size_t)in.P.n_binned_theta * in.P.n_binned_phi : (size_t)d.n_view;
    const size_t n_nu = (size_t)d.n_nu;
    Cubes C;
    std::vector<double> nu;     // exact frequencies of the group (monochromatic)
    if (frequencies) nu.assign(frequencies->begin() + (d.inu_min - 1), frequencies->begin() + d.inu_max);
    double norm = 1.0;          // with filters the flux stays F_nu dnu: the curve carries the normalisation (:644-651)
    if (!frequencies && !d.use_filters) {
        const double r = d.nu_max / d.nu_min, n = (double)d.n_nu;
        norm = std::pow(r, 0.5 / n) - std::pow(r, -0.5 / n);
    }
    auto get = [&](int which, size_t n) {
        std::vector<double> a(n);
        uint64_t nn = n;
        check(hyp_peeled_get(h, (int)ig, which, a.data(), &nn), h);
        return a;
    };
    auto scale = [&](std::vector<double> &a, bool root) {
        for (size_t i = 0; i < a.size(); i++) {
            double v = root ? std::sqrt(a[i]) : a[i];
            a[i] = frequencies ? v * nu[i % n_nu] : v / norm;
        }
    };
    if (d.compute_sed) {
        const size_t n_ap = (size_t)d.n_ap, n = ns * n_orig * n_view * n_ap * n_nu;
        C.sed_dims = {ns, (hsize_t)n_orig, n_view, n_ap, n_nu};
        C.seds = get(0, n);
        scale(C.seds, false);
        for (size_t o = 0; o < ns * n_orig * n_view; o++)
            for (size_t ia = 1; ia < n_ap; ia++)
                for (size_t k = 0; k < n_nu; k++) C.seds[(o * n_ap + ia) * n_nu + k] += C.seds[(o * n_ap + ia - 1) * n_nu + k];
        if (d.uncertainties) {
            C.seds_unc = get(1, n);
            scale(C.seds_unc, true);
            for (size_t o = 0; o < ns * n_orig * n_view; o++) {
                for (size_t k = 0; k < n_nu; k++) {
                    double acc = 0.0;
                    for (size_t ia = 0; ia < n_ap; ia++) {
                        double &u = C.seds_unc[(o * n_ap + ia) * n_nu + k];
                        acc += u * u; u = std::sqrt(acc);
                    }
                }
            }
        }
    }
    if (d.compute_image) {
        const size_t n = ns * n_orig * n_view * (size_t)d.n_y * (size_t)d.n_x * n_nu;
        C.img_dims = {ns, (hsize_t)n_orig, n_view, (hsize_t)d.n_y, (hsize_t)d.n_x, n_nu};
        C.images = get(2, n);
        scale(C.images, false);
        if (d.uncertainties) { C.images_unc = get(3, n); scale(C.images_unc, true); }
    }
    return C;
}

void write_image_datasets(hid_t g, const Input &in, const Group &G, const hyp_peeled_desc &d, const Cubes &C, bool numinmax)
{
    const int kind = G.io_bytes == 4 ? 4 : 8;      // image_type.f90:690-700
    for (int which = 0; which < 2; which++) {
        const std::vector<double> &a = which == 0 ? C.seds : C.images, &u = which == 0 ? C.seds_unc : C.images_unc;
        const std::vector<hsize_t> &dims = which == 0 ? C.sed_dims : C.img_dims;
        if (dims.empty()) continue;
        const char *name = which == 0 ? "seds" : "images";
        Hid ds(put_dataset(g, name, dims, a.data(), kind), 2);
        if (numinmax) { put_attr_dbl(ds, "numin", d.nu_min); put_attr_dbl(ds, "numax", d.nu_max); }     // :701-706
        if (which == 0) { put_attr_dbl(ds, "apmin", d.ap_min); put_attr_dbl(ds, "apmax", d.ap_max); }
        else { put_attr_dbl(ds, "xmin", d.x_min); put_attr_dbl(ds, "xmax", d.x_max); put_attr_dbl(ds, "ymin", d.y_min); put_attr_dbl(ds, "ymax", d.y_max); }
        put_attr_str(ds, "track_origin", G.track_origin);
        if (d.track_origin == 2) { put_attr_i32(ds, "n_sources", in.P.n_sources); put_attr_i32(ds, "n_dust", in.P.n_dust); }
        else if (d.track_origin == 3) put_attr_i32(ds, "track_n_scat", d.track_n_scat);
        if (!u.empty()) { Hid du(put_dataset(g, which == 0 ? "seds_unc" : "images_unc", dims, u.data(), kind), 2); }
    }
}

herr_t copy_attr_cb(hid_t loc, const char *name, const H5A_info_t *, void *op)
{
    hid_t dst = *(hid_t *)op;
    Hid a(H5Aopen(loc, name, H5P_DEFAULT), 3);
    Hid t(H5Aget_type(a), 4);
    Hid s(H5Aget_space(a), 5);
    const size_t n = (size_t)std::max<hssize_t>(H5Sget_simple_extent_npoints(s), 1) * H5Tget_size(t);
    std::vector<char> buf(n + 16);
    if (H5Aread(a, t, buf.data()) < 0) return -1;
    Hid b(H5Acreate2(dst, name, t, s, H5P_DEFAULT, H5P_DEFAULT), 3);
    if (b < 0 || H5Awrite(b, t, buf.data()) < 0) return -1;
    if (H5Tis_variable_str(t) > 0) H5Dvlen_reclaim(t, s, H5P_DEFAULT, buf.data());
    return 0;
}

// one grid quantity of an iteration: write_grid_3d / write_grid_4d (src/grid/grid_io.f90, grid_io_amr_template.f90)
void write_quantity(hid_t g, const Input &in, const char *name, const std::vector<double> &a, const std::vector<hsize_t> &lead, int kind)
{
    if (in.grid_type != "amr") {
        std::vector<hsize_t> dims(lead);
        dims.insert(dims.end(), in.cell_dims.begin(), in.cell_dims.end());
        Hid d(put_dataset(g, name, dims, a.data(), kind), 2);
        put_attr_str(d, "geometry", in.geometry_id);
        return;
    }
    size_t planes = 1;
    for (hsize_t x : lead) planes *= (size_t)x;
    size_t start = 0;
    for (size_t k = 0; k < in.amr_paths.size(); k++) {
        const int32_t *n = in.P.grid.amr_n + 3 * k;
        const size_t nc = (size_t)n[0] * n[1] * n[2];
        std::vector<double> part(planes * nc);
        for (size_t p = 0; p < planes; p++) std::copy(a.begin() + (long)(p * in.n_cells + start), a.begin() + (long)(p * in.n_cells + start + nc), part.begin() + (long)(p * nc));
        const std::string lv = in.amr_paths[k].substr(0, in.amr_paths[k].find('/'));
        if (!has_link(g, lv.c_str())) { Hid t(H5Gcreate2(g, lv.c_str(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), 1); }
        if (!has_link(g, in.amr_paths[k].c_str())) { Hid t(H5Gcreate2(g, in.amr_paths[k].c_str(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), 1); }
        Hid gg(H5Gopen2(g, in.amr_paths[k].c_str(), H5P_DEFAULT), 1);
        std::vector<hsize_t> dims(lead);
        dims.push_back((hsize_t)n[2]); dims.push_back((hsize_t)n[1]); dims.push_back((hsize_t)n[0]);
        Hid d(put_dataset(gg, name, dims, part.data(), kind), 2);
        start += nc;
    }
}

int run(const char *input, const char *output, bool overwrite, Comm &comm)
{
    const std::string date_started = now_string();
    const double t0 = (double)clock() / CLOCKS_PER_SEC;
    struct timespec w0; clock_gettime(CLOCK_MONOTONIC, &w0);
    printf(" %s\n Hyperion-AMD native driver (C ABI v%d)\n Started on %s\n Input:  %s\n Output: %s\n %s\n", std::string(60, '-').c_str(),
           hyp_abi_version(), date_started.c_str(), input, output, std::string(60, '-').c_str());
    if (access(input, R_OK) != 0) throw Fail(fmt("File does not exist: %s", input));
    if (comm.main_process() && access(output, F_OK) == 0) {
        if (!overwrite) throw Fail(fmt("Output file %s already exists (use -f)", output));
        char *ai = realpath(input, nullptr), *ao = realpath(output, nullptr);
        const bool same = ai && ao && strcmp(ai, ao) == 0;
        free(ai); free(ao);
        if (same) throw Fail("input and output are the same file");
    }
    // the input is read and validated BEFORE an existing output is removed: a bad .rtin must not cost the previous result
    Input in;
    read_rtin(input, in);
    if (comm.main_process() && access(output, F_OK) == 0) unlink(output);
    const hyp_config &cfg = in.P.config;

    // the output exists from the start, date_ended is written last: its presence marks success (main.f90:130-136,338-344).
    // Only rank 0 has a file: the other ranks write the same groups into memory (HDF5 core driver, no backing store).
    Hid fapl(H5Pcreate(H5P_FILE_ACCESS), 6);
    if (!comm.main_process()) H5Pset_fapl_core(fapl, (size_t)1 << 24, 0);
    Hid fo(H5Fcreate(comm.main_process() ? output : fmt("hyperion_amd_rank%d.mem", comm.rank).c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, fapl), 0);
    if (fo < 0) throw Fail(fmt("cannot create output file %s", output));
    Hid root(H5Gopen2(fo, "/", H5P_DEFAULT), 1);
    put_attr_str(root, "date_started", date_started);
    put_attr_str(root, "fortran_version", FORTRAN_VERSION);
    {
        char *abs = realpath(input, nullptr);
        const std::string ap = abs ? abs : input;
        free(abs);
        if (in.copy_input) {        // main.f90:138-150
            Hid fi(H5Fopen(input, H5F_ACC_RDONLY, H5P_DEFAULT), 0);
            Hid gi(H5Gcreate2(fo, "Input", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), 1);
            Hid ri(H5Gopen2(fi, "/", H5P_DEFAULT), 1);
            for (const std::string &k : children(ri))
                if (H5Ocopy(fi, k.c_str(), gi, k.c_str(), H5P_DEFAULT, H5P_DEFAULT) < 0) throw Fail("cannot copy the input into the output");
            hid_t dst = gi;
            if (H5Aiterate2(ri, H5_INDEX_NAME, H5_ITER_INC, nullptr, copy_attr_cb, &dst) < 0) throw Fail("cannot copy the input attributes");
        } else if (H5Lcreate_external(ap.c_str(), "/", fo, "Input", H5P_DEFAULT, H5P_DEFAULT) < 0) throw Fail("cannot link the input");
    }
    H5Fflush(fo, H5F_SCOPE_GLOBAL);

    hyp_handle h = nullptr;
    // one process per GPU: HYP_DEVICE, else the launcher's local rank (torchrun / Open MPI / Slurm), else the rank modulo the
    // devices of the node, else device 0
    int device = comm.local_rank >= 0 ? comm.local_rank : 0;
    if (comm.local_rank < 0 && comm.on) {
        int nd = 0;
        if (hipGetDeviceCount(&nd) == hipSuccess && nd > 0) device = comm.rank % nd;
    }
    comm.init(output, device);
    if (comm.on) printf(" [mpi] rank %d of %d on device %d, RCCL all-reduce of the accumulator block per iteration\n", comm.rank, comm.size, device);
    if (hyp_create(&in.P, device, &h) != 0) { const char *m = hyp_last_error(nullptr); throw Fail(m && *m ? m : "hyp_create failed"); }
    printf(" [main] using random seed = %lld\n", (long long)cfg.seed);
    const size_t plane = (size_t)in.P.n_dust * in.n_cells;
    const int pkind = in.physics_io_bytes == 4 ? 4 : 8;

    // ---- Lucy iterations: main.f90:167-234 ----
    bool converged = false;
    long long n_done = in.n_initial_iter;
    double value_prev = std::numeric_limits<double>::infinity();
    for (long long it = 1; it <= in.n_initial_iter; it++) {
        printf(" [main] starting Lucy iteration %lld\n", it);
        Iteration rec;
        rec.index = it;
        std::vector<double> se(plane);
        hyp_iter_stats st{};
        lucy_iteration(h, comm, (uint64_t)in.n_initial_photons, (int)it, se.data(), &st);
        if (cfg.count_photons) {
            int64_t inexact = 0;
            if (hyp_get_option(h, "n_photons_inexact", &inexact) == 0 && inexact)
                printf(" [main] WARNING: n_photons of iteration %lld is an upper bound (a packet overflowed its visited-cell set)\n", it);
        }
        printf(" [main] exiting Lucy iteration\n");
        rec.killed_geo = st.killed_geo; rec.killed_int = st.killed_int;
        if (in.check_convergence) {     // specific_energy_converged: grid_physics_3d.f90:637-689
            double value = 0.0; int status = 0;
            check(hyp_convergence_value(h, in.conv_pct, &value, &status), h);
            converged = false;
            if (status == 2) printf(" [specific_energy_converged] could not check for convergence, as the only cells that changed had zero value before or after\n");
            else if (status != 3) {
                if (value_prev < std::numeric_limits<double>::infinity()) {
                    if (value == 0.0) converged = true;
                    else {
                        const double ratio = std::max(value_prev / value, value / value_prev);
                        converged = value < in.conv_abs && std::fabs(ratio) < in.conv_rel;
                    }
                }
                value_prev = value;
            }
            if (converged) printf("      ------ Specific energy calculation converged -----\n");
        }
        const long long last = (in.check_convergence && converged) ? it : in.n_initial_iter;
        // output_grid: grid_generic.f90:29-130
        Hid g(H5Gcreate2(fo, fmt("iteration_%05lld", it).c_str(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), 1);
        put_attr_i32(g, "killed_photons_geo", (int32_t)rec.killed_geo);
        put_attr_i32(g, "killed_photons_int", (int32_t)rec.killed_int);
        if (want(in.out_n_photons, it, last)) {
            std::vector<double> a(in.n_cells);
            check(hyp_get_n_photons(h, a.data()), h);
            write_quantity(g, in, "n_photons", a, {}, -8);
        }
        if (want(in.out_specific_energy, it, last)) write_quantity(g, in, "specific_energy", se, {(hsize_t)in.P.n_dust}, pkind);
        if (want(in.out_spectrum, it, last)) {        // grid_generic.f90:71-93
            const size_t nb = (size_t)cfg.n_spectrum_bins;
            std::vector<double> a(nb * plane), e(nb + 1);
            check(hyp_get_specific_energy_spectrum(h, a.data(), e.data()), h);
            Hid de(put_dataset(g, "specific_energy_spectrum_bin_edges", {(hsize_t)(nb + 1)}, e.data(), 8), 2);
            write_quantity(g, in, "specific_energy_spectrum", a, {(hsize_t)nb, (hsize_t)in.P.n_dust}, pkind);
        }
        if (want(in.out_density, it, last) || want(in.out_density_diff, it, last)) {
            std::vector<double> a(plane);
            check(hyp_get_density(h, a.data()), h);
            if (want(in.out_density, it, last)) write_quantity(g, in, "density", a, {(hsize_t)in.P.n_dust}, pkind);
            if (want(in.out_density_diff, it, last)) {
                for (size_t i = 0; i < plane; i++) a[i] -= in.density[i];
                write_quantity(g, in, "density_diff", a, {(hsize_t)in.P.n_dust}, pkind);
            }
        }
        if (in.check_convergence && converged) { n_done = it; break; }
    }
    put_attr_str(root, "converged", converged ? "yes" : "no");
    put_attr_i32(root, "iterations", (int32_t)n_done);

    // ---- final iteration(s): main.f90:236-305 ----
    hyp_iter_stats fst, rst;
    std::memset(&fst, 0, sizeof fst); std::memset(&rst, 0, sizeof rst);
    std::vector<double> freq;
    if (cfg.monochromatic) freq.assign(cfg.frequencies, cfg.frequencies + cfg.n_frequencies);
    printf(" [main] starting final iteration\n");
    if (cfg.monochromatic) { if (in.P.n_peeled > 0) mono_iteration(h, comm, (uint64_t)in.n_last_photons_sources, (uint64_t)in.n_last_photons_dust, cfg.n_frequencies, &fst); }
    else if (in.n_last_photons > 0) final_iteration(h, comm, (uint64_t)in.n_last_photons, &fst);
    else printf("      ------------------ Skipping ------------------\n");
    printf(" [main] exiting final iteration\n");
    if (cfg.raytracing) {
        printf(" [main] starting raytracing iteration\n");
        raytracing_iteration(h, comm, (uint64_t)in.n_ray_photons_sources, (uint64_t)in.n_ray_photons_dust, &rst);
        printf(" [main] exiting raytracing iteration\n");
    }
    const bool have_cubes = !(cfg.monochromatic && in.P.n_peeled == 0);
    if (in.P.n_peeled > 0 && have_cubes) {
        Hid gp(H5Gcreate2(fo, "Peeled", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), 1);
        for (size_t ig = 0; ig < (size_t)in.P.n_peeled; ig++) {
            const Group &G = in.groups[ig];
            const hyp_peeled_desc &d = in.groups_desc[ig];
            Cubes C = fetch_group(h, in, ig, cfg.monochromatic ? &freq : nullptr);
            Hid g(H5Gcreate2(gp, fmt("group_%05zu", ig + 1).c_str(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), 1);
            put_attr_str(g, "inside_observer", d.inside_observer ? "yes" : "no");
            put_attr_dbl(g, "d_min", d.d_min); put_attr_dbl(g, "d_max", d.d_max);
            write_image_datasets(g, in, G, d, C, !cfg.monochromatic && !d.use_filters);
            if (d.use_filters) {        // image_type.f90:773-777
                put_attr_str(g, "use_filters", "yes");
                put_attr_i32(g, "n_filt", d.n_nu);
                Hid df(put_dataset(g, "filt_nu0", {(hsize_t)G.filt_nu0.size()}, G.filt_nu0.data(), 8), 2);
            }
            if (cfg.monochromatic) {    // image_type.f90:781-784: table `frequencies`, column nu
                const hsize_t n = (hsize_t)(d.inu_max - d.inu_min + 1);
                Hid t(H5Tcreate(H5T_COMPOUND, sizeof(double)), 4);
                H5Tinsert(t, "nu", 0, H5T_NATIVE_DOUBLE);
                Hid s(H5Screate_simple(1, &n, nullptr), 5);
                Hid dd(H5Dcreate2(g, "frequencies", t, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), 2);
                if (dd < 0 || H5Dwrite(dd, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, freq.data() + (d.inu_min - 1)) < 0) throw Fail("cannot write the frequencies");
            }
        }
    }
    if (in.P.binned && !cfg.monochromatic) {       // binned_images_write (images_binned.f90:89-93): image_write into /Binned
        const size_t ig = (size_t)in.P.n_peeled;
        Cubes C = fetch_group(h, in, ig, nullptr);
        Hid g(H5Gcreate2(fo, "Binned", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), 1);
        Group G = in.groups[ig];
        G.io_bytes = 8;
        write_image_datasets(g, in, G, in.groups_desc[ig], C, true);
    }
    hyp_destroy(h);
    comm.finalize();
    put_attr_i32(root, "killed_photons_geo_final", (int32_t)fst.killed_geo);
    put_attr_i32(root, "killed_photons_int_final", (int32_t)fst.killed_int);
    put_attr_i32(root, "killed_photons_geo_raytracing", (int32_t)rst.killed_geo);
    put_attr_i32(root, "killed_photons_int_raytracing", (int32_t)rst.killed_int);
    struct timespec w1; clock_gettime(CLOCK_MONOTONIC, &w1);
    const double wall = (double)(w1.tv_sec - w0.tv_sec) + 1e-9 * (double)(w1.tv_nsec - w0.tv_nsec);
    (void)t0;
    put_attr_dbl(root, "cpu_time", wall);
    put_attr_str(root, "date_ended", now_string());      // last: its presence marks success
    printf(" Total time elapsed: %16.2f\n", wall);
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    // main.f90:74-106: [-f] input_file output_file; --check-input parses the input and stops (no GPU needed)
    bool overwrite = false, check_only = false;
    int n_ranks = 0;
    std::vector<const char *> pos;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "-f")) overwrite = true;
        else if (!std::strcmp(argv[i], "--check-input")) check_only = true;
        else if (!std::strcmp(argv[i], "--ranks") && i + 1 < argc) n_ranks = atoi(argv[++i]);
        else pos.push_back(argv[i]);
    }
    if ((check_only && pos.size() != 1) || (!check_only && pos.size() != 2) || n_ranks < 0) {
        fprintf(stderr, "Usage: %s [-f] [--ranks N] input_file output_file\n", argv[0]);
        return 2;
    }
    // --ranks N: be the launcher -- N ranks of this executable on this node, one GPU each (what `mpirun -n N hyperion_car_mpi`
    // is to the reference, scripts/hyperion:62-92), started before anything touches HIP or HDF5
    std::vector<pid_t> children;
    if (n_ranks > 0 && !check_only) {
        setenv("WORLD_SIZE", std::to_string(n_ranks).c_str(), 1);
        setenv("HYP_LAUNCHER_PID", std::to_string((long long)getpid()).c_str(), 1);
        int my_rank = 0;
        for (int r = 1; r < n_ranks; r++) {
            pid_t pid = fork();
            if (pid < 0) { fprintf(stderr, " ERROR: cannot start rank %d\n", r); return 1; }
            if (pid == 0) { my_rank = r; children.clear(); break; }
            children.push_back(pid);
        }
        setenv("RANK", std::to_string(my_rank).c_str(), 1);
        setenv("LOCAL_RANK", std::to_string(my_rank).c_str(), 1);
    }
    H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);      // errors are reported through Fail, not HDF5's stack dump
    try {
        if (check_only) {
            Input in;
            read_rtin(pos[0], in);
            printf("grid_type %s n_cells %zu n_dust %d n_sources %d n_peeled %d binned %d n_initial_iter %lld n_initial_photons %lld n_last_photons %lld\n",
                   in.grid_type.c_str(), in.n_cells, in.P.n_dust, in.P.n_sources, in.P.n_peeled, in.P.binned ? 1 : 0, in.n_initial_iter,
                   in.n_initial_photons, in.n_last_photons);
            double s = 0.0;
            for (double v : in.density) s += v;
            printf("density_sum %.17g seed %lld pda %d mrw %d raytracing %d monochromatic %d output_specific_energy %s physics_io_bytes %d\n", s,
                   (long long)in.P.config.seed, in.P.config.pda, in.P.config.mrw, in.P.config.raytracing, in.P.config.monochromatic,
                   in.out_specific_energy.c_str(), in.physics_io_bytes);
            for (size_t i = 0; i < in.dust.size(); i++)
                printf("dust %zu n_nu %d n_mu %d n_jnu %d n_enu %d n_e %d sublimation %d chi0 %.17g\n", i, in.dust[i].n_nu, in.dust[i].n_mu, in.dust[i].n_jnu,
                       in.dust[i].n_enu, in.dust[i].n_e, in.dust[i].sublimation_mode, in.dust[i].chi[0]);
            uint64_t dg[4];
            if (hyp_problem_digest(&in.P, dg) == 0)
                printf("digest %016llx %016llx %016llx %016llx\n", (unsigned long long)dg[0], (unsigned long long)dg[1], (unsigned long long)dg[2], (unsigned long long)dg[3]);
            for (size_t i = 0; i < in.groups.size(); i++)
                printf("group %zu n_view %d n_nu %d n_x %d n_y %d n_ap %d track_origin %d uncertainties %d stokes %d filters %d io_bytes %d\n", i, in.groups_desc[i].n_view,
                       in.groups_desc[i].n_nu, in.groups_desc[i].n_x, in.groups_desc[i].n_y, in.groups_desc[i].n_ap, in.groups_desc[i].track_origin,
                       in.groups_desc[i].uncertainties, in.groups_desc[i].compute_stokes, in.groups_desc[i].use_filters, in.groups[i].io_bytes);
            return 0;
        }
    } catch (const std::exception &e) {
        fprintf(stderr, " ERROR: %s\n", e.what());
        return 1;
    }
    Comm comm;
    int rc = 0;
    try {
        const std::string exe = argv[0];
        comm.from_env(pos[1], exe.size() >= 4 && exe.compare(exe.size() - 4, 4, "_mpi") == 0);
        rc = run(pos[0], pos[1], overwrite, comm);
    } catch (const std::exception &e) {
        // the reference's error(): message on stderr, the output (if any) has no date_ended
        comm.abort_others(e.what());
        fprintf(stderr, " ERROR: %s\n", e.what());
        fprintf(stderr, "An error occurred, and the run did not complete\n");
        rc = 1;
    }
    for (pid_t pid : children) {        // --ranks: the launcher's status is the worst of its ranks
        int st = 0;
        if (waitpid(pid, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = rc ? rc : 1;
    }
    return rc;
}
