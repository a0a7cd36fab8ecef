/* fake_libnetcdf.c -- a TEST DOUBLE of the netcdf-c C API, compiled by tests/test_netcdf_loader.py into a shared object
 * that the loader's run-time binding (CRF_LIBNETCDF -> dlopen) picks up.  This image has no libnetcdf / libhdf5 and no
 * NetCDF-4 file; the double lets the tests drive the loader's library back end -- the call sequence nc_open, nc_inq,
 * nc_inq_dim, nc_inq_var, nc_inq_att, nc_get_att_text/float, nc_get_vara_float, nc_close with the signatures and
 * nc_type codes of netcdf.h -- with a data set described by a small side-car text file "<path>.fake":
 *     dims   N  name len  name len ...
 *     var    name type(5=float,6=double) rank dimid... [standard_name=TEXT] [fill=VALUE]
 *     ... followed by the variable's values as text (row-major), one variable after the other.
 * It proves the binding, the conversions and the conventions applied on top; it does NOT prove HDF5 decoding, which is
 * the real library's job. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXD 8
#define MAXV 8
typedef struct { char name[64]; size_t len; } Dim;
typedef struct { char name[64]; int type, rank, dimids[4]; char standard_name[64]; int has_fill; float fill; double* data; size_t n; } Var;
static Dim g_dims[MAXD];
static Var g_vars[MAXV];
static int g_ndims, g_nvars, g_open;

int nc_open(const char* path, int mode, int* ncidp) {
    char side[4096];
    (void)mode;
    snprintf(side, sizeof side, "%s.fake", path);
    FILE* f = fopen(side, "r");
    if (!f) return -31; /* "NC_ENOTNC"-like */
    char word[64];
    g_ndims = g_nvars = 0;
    while (fscanf(f, "%63s", word) == 1) {
        if (!strcmp(word, "dims")) {
            if (fscanf(f, "%d", &g_ndims) != 1) return -1;
            for (int d = 0; d < g_ndims; d++)
                if (fscanf(f, "%63s %zu", g_dims[d].name, &g_dims[d].len) != 2) return -1;
        } else if (!strcmp(word, "var")) {
            Var* v = &g_vars[g_nvars++];
            memset(v, 0, sizeof *v);
            if (fscanf(f, "%63s %d %d", v->name, &v->type, &v->rank) != 3) return -1;
            v->n = 1;
            for (int r = 0; r < v->rank; r++) {
                if (fscanf(f, "%d", &v->dimids[r]) != 1) return -1;
                v->n *= g_dims[v->dimids[r]].len;
            }
            char opt[128];
            long pos = ftell(f);
            while (fscanf(f, "%127s", opt) == 1) {
                if (!strncmp(opt, "standard_name=", 14)) {
                    strncpy(v->standard_name, opt + 14, 63);
                } else if (!strncmp(opt, "fill=", 5)) {
                    v->has_fill = 1;
                    v->fill = (float)atof(opt + 5);
                } else {
                    fseek(f, pos, SEEK_SET);
                    break;
                }
                pos = ftell(f);
            }
            v->data = (double*)malloc(v->n * sizeof(double));
            for (size_t i = 0; i < v->n; i++)
                if (fscanf(f, "%lf", &v->data[i]) != 1) return -1;
        }
    }
    fclose(f);
    g_open = 1;
    *ncidp = 65536;
    return 0;
}
int nc_close(int ncid) {
    (void)ncid;
    for (int i = 0; i < g_nvars; i++) free(g_vars[i].data);
    g_open = 0;
    return 0;
}
int nc_inq(int ncid, int* ndimsp, int* nvarsp, int* ngattsp, int* unlimdimidp) {
    (void)ncid;
    *ndimsp = g_ndims;
    *nvarsp = g_nvars;
    *ngattsp = 0;
    *unlimdimidp = -1;
    return 0;
}
int nc_inq_dim(int ncid, int dimid, char* name, size_t* lenp) {
    (void)ncid;
    if (dimid < 0 || dimid >= g_ndims) return -46;
    strcpy(name, g_dims[dimid].name);
    *lenp = g_dims[dimid].len;
    return 0;
}
int nc_inq_var(int ncid, int varid, char* name, int* xtypep, int* ndimsp, int* dimidsp, int* nattsp) {
    (void)ncid;
    if (varid < 0 || varid >= g_nvars) return -49;
    const Var* v = &g_vars[varid];
    strcpy(name, v->name);
    *xtypep = v->type;
    *ndimsp = v->rank;
    for (int r = 0; r < v->rank; r++) dimidsp[r] = v->dimids[r];
    *nattsp = (v->standard_name[0] != 0) + v->has_fill;
    return 0;
}
int nc_inq_att(int ncid, int varid, const char* name, int* xtypep, size_t* lenp) {
    (void)ncid;
    const Var* v = &g_vars[varid];
    if (!strcmp(name, "standard_name") && v->standard_name[0]) {
        *xtypep = 2; /* NC_CHAR */
        *lenp = strlen(v->standard_name);
        return 0;
    }
    if (!strcmp(name, "_FillValue") && v->has_fill) {
        *xtypep = v->type;
        *lenp = 1;
        return 0;
    }
    return -43; /* NC_ENOTATT */
}
int nc_get_att_text(int ncid, int varid, const char* name, char* out) {
    (void)ncid;
    (void)name;
    memcpy(out, g_vars[varid].standard_name, strlen(g_vars[varid].standard_name)); /* no terminator, like netcdf-c */
    return 0;
}
int nc_get_att_float(int ncid, int varid, const char* name, float* out) {
    (void)ncid;
    (void)name;
    *out = g_vars[varid].fill;
    return 0;
}
int nc_get_vara_float(int ncid, int varid, const size_t* start, const size_t* count, float* out) {
    (void)ncid;
    const Var* v = &g_vars[varid];
    size_t stride[4], total = 1;
    for (int r = v->rank - 1; r >= 0; r--) {
        stride[r] = total;
        total *= g_dims[v->dimids[r]].len;
    }
    for (int r = 0; r < v->rank; r++)
        if (start[r] + count[r] > g_dims[v->dimids[r]].len) return -57; /* NC_EEDGE */
    size_t idx[4] = {0, 0, 0, 0}, o = 0;
    for (;;) {
        size_t off = 0;
        for (int r = 0; r < v->rank; r++) off += (start[r] + idx[r]) * stride[r];
        out[o++] = (float)v->data[off];
        int r = v->rank - 1;
        while (r >= 0 && ++idx[r] == count[r]) idx[r--] = 0;
        if (r < 0) break;
    }
    return 0;
}
const char* nc_strerror(int status) {
    static char buf[64];
    snprintf(buf, sizeof buf, "fake netcdf status %d", status);
    return buf;
}
