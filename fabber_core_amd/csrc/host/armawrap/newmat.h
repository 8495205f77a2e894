/*
 * armawrap/newmat.h - minimal NEWMAT-compatible dense matrix value types.
 *
 * fabber's plugin surface (FwdModel::EvaluateModel, MVNDist::means, FabberRunData::GetVoxelData,
 * ...) is written against the NEWMAT API, which the reference gets from FSL's "armawrap"
 * (Armadillo-backed; not part of the reference tree). This header is an independent, small
 * implementation of the subset those signatures and typical model sources use, so that model
 * plugins compile unchanged against this library. It is host-side glue only: none of the
 * per-voxel numerics run through it (those live in the HIP kernels).
 *
 * Supported: Matrix, ColumnVector, RowVector, SymmetricMatrix, DiagonalMatrix, IdentityMatrix,
 * ReturnMatrix; 1-based element access; ReSize; scalar fill; "<<" list fill and lossy assign;
 * Row/Column/Rows/Columns/SubMatrix/SymSubMatrix as assignable views; t(), i(),
 * LogDeterminant(), Trace, Sum, Maximum, Minimum, MaximumAbsoluteValue, AsScalar, AsRow,
 * AsColumn, IsZero; + - * / with matrices and scalars; & (stack) and | (append); ==, !=;
 * stream output.
 */
#pragma once

#include <algorithm>
#include <cmath>
#include <exception>
#include <iomanip>
#include <memory>
#include <mutex>
#include <new>
#include <stdlib.h>
#include <sys/mman.h>
#include <ostream>
#include <string>
#include <utility>
#include <vector>

namespace NEWMAT
{
typedef double Real;

class Exception : public std::exception
{
public:
    explicit Exception(const std::string &msg = "NEWMAT exception")
        : m_msg(msg)
    {
    }
    virtual ~Exception() throw()
    {
    }
    virtual const char *what() const throw()
    {
        return m_msg.c_str();
    }

private:
    std::string m_msg;
};
class SingularException : public Exception
{
public:
    SingularException()
        : Exception("matrix is singular")
    {
    }
};
class IndexException : public Exception
{
public:
    IndexException()
        : Exception("index out of range")
    {
    }
};
class IncompatibleDimensionsException : public Exception
{
public:
    IncompatibleDimensionsException()
        : Exception("incompatible dimensions")
    {
    }
};

class LogAndSign
{
public:
    LogAndSign(Real logv = 0, int sign = 1)
        : m_log(logv)
        , m_sign(sign)
    {
    }
    Real LogValue() const
    {
        return m_log;
    }
    int Sign() const
    {
        return m_sign;
    }
    Real Value() const
    {
        return m_sign * std::exp(m_log);
    }

private:
    Real m_log;
    int m_sign;
};

class Matrix;

// Assignable rectangular view into a Matrix (what Row(), Column(), SubMatrix()... return)
class MatrixView
{
public:
    MatrixView(Matrix *m, int r0, int c0, int nr, int nc)
        : m_m(m)
        , m_r0(r0)
        , m_c0(c0)
        , m_nr(nr)
        , m_nc(nc)
    {
    }
    inline MatrixView &operator=(const Matrix &src);
    inline MatrixView &operator=(const MatrixView &src);
    inline MatrixView &operator=(Real v);
    inline MatrixView &operator<<(const Matrix &src);
    inline operator Matrix() const;
    inline Matrix AsMatrix() const;
    int Nrows() const
    {
        return m_nr;
    }
    int Ncols() const
    {
        return m_nc;
    }
    inline Real operator()(int i, int j) const;
    inline Real operator()(int i) const;
    // frequently chained operations
    inline Matrix t() const;
    inline Real Sum() const;
    inline Real Maximum() const;
    inline Real Minimum() const;
    inline Real AsScalar() const;
    inline Matrix Rows(int a, int b) const;
    inline Matrix Columns(int a, int b) const;
    inline bool IsZero() const;

private:
    Matrix *m_m;
    int m_r0, m_c0, m_nr, m_nc;
};

// Helper for "m << a << b << c;" element-list fill (row-major)
class ListFiller
{
public:
    ListFiller(Matrix *m, int pos)
        : m_m(m)
        , m_pos(pos)
    {
    }
    inline ListFiller operator<<(Real v);

private:
    Matrix *m_m;
    int m_pos;
};

// Big blocks (volumes: tens to hundreds of MB) are not given back to the system when an image dies but kept for the
// next one of the same size: a run through the C ABI allocates ~1 GB of images between fabber_new and fabber_destroy, and
// mapping, first-touching and unmapping them was a third of a million-voxel run (35 ms in fabber_destroy alone). The
// cache is bounded (FVB_HOST_CACHE_BYTES, default 3 GiB; 0 = off), process-wide, and emptied by trim().
class BigBlockCache
{
public:
    static BigBlockCache &instance()
    {
        static BigBlockCache cache;
        return cache;
    }
    void *take(std::size_t bytes)
    {
        std::lock_guard<std::mutex> lock(m_mu);
        for (std::size_t i = 0; i < m_free.size(); i++)
            if (m_free[i].second == bytes)
            {
                void *p = m_free[i].first;
                m_free[i] = m_free.back();
                m_free.pop_back();
                m_held -= bytes;
                return p;
            }
        return nullptr;
    }
    bool give(void *p, std::size_t bytes)
    {
        std::lock_guard<std::mutex> lock(m_mu);
        if (m_held + bytes > m_cap)
            return false;
        m_free.push_back(std::make_pair(p, bytes));
        m_held += bytes;
        return true;
    }
    /** give back everything beyond `keep` bytes (largest blocks first) */
    void trim(std::size_t keep)
    {
        std::lock_guard<std::mutex> lock(m_mu);
        while (m_held > keep && !m_free.empty())
        {
            std::size_t big = 0;
            for (std::size_t i = 1; i < m_free.size(); i++)
                if (m_free[i].second > m_free[big].second)
                    big = i;
            free(m_free[big].first);
            m_held -= m_free[big].second;
            m_free[big] = m_free.back();
            m_free.pop_back();
        }
    }
    std::size_t held()
    {
        std::lock_guard<std::mutex> lock(m_mu);
        return m_held;
    }

private:
    BigBlockCache()
        : m_held(0)
        , m_cap(std::size_t(3) << 30)
    {
        if (const char *e = getenv("FVB_HOST_CACHE_BYTES"))
            m_cap = (std::size_t)strtoull(e, nullptr, 10);
    }
    ~BigBlockCache()
    {
        for (std::size_t i = 0; i < m_free.size(); i++)
            free(m_free[i].first);
    }
    std::mutex m_mu;
    std::vector<std::pair<void *, std::size_t> > m_free;
    std::size_t m_held, m_cap;
};

// Storage that can be sized WITHOUT being written: value-initialising the 168 MB result image of a million voxels (or
// the 800 MB of a series) touches every page from one thread before the real contents overwrite them.
template <class T>
struct DefaultInitAllocator : std::allocator<T>
{
    template <class U>
    struct rebind
    {
        typedef DefaultInitAllocator<U> other;
    };
    DefaultInitAllocator() = default;
    template <class U>
    DefaultInitAllocator(const DefaultInitAllocator<U> &)
    {
    }
    // Large blocks come 2 MB-aligned with a request for transparent huge pages: a freshly allocated volume is
    // faulted in - and given back at the end of the run - in 2 MB steps instead of 4 KB ones (first touch and
    // munmap of ~1 GB of images were a quarter of a second of a million-voxel run through the C ABI).
    static constexpr std::size_t BIG = std::size_t(1) << 20, HUGE_PAGE = std::size_t(2) << 20;
    T *allocate(std::size_t n)
    {
        const std::size_t bytes = n * sizeof(T);
        if (bytes < BIG)
            return std::allocator<T>::allocate(n);
        const std::size_t rounded = (bytes + HUGE_PAGE - 1) & ~(HUGE_PAGE - 1);
        void *p = BigBlockCache::instance().take(rounded);
        if (p)
            return static_cast<T *>(p);
        if (posix_memalign(&p, HUGE_PAGE, rounded) != 0)
            throw std::bad_alloc();
#ifdef MADV_HUGEPAGE
        (void)madvise(p, rounded, MADV_HUGEPAGE);
#endif
        return static_cast<T *>(p);
    }
    void deallocate(T *p, std::size_t n)
    {
        if (n * sizeof(T) < BIG)
            std::allocator<T>::deallocate(p, n);
        else
        {
            const std::size_t rounded = (n * sizeof(T) + HUGE_PAGE - 1) & ~(HUGE_PAGE - 1);
            if (!BigBlockCache::instance().give(p, rounded))
                free(p);
        }
    }
    template <class U>
    void construct(U *p)
    {
        ::new ((void *)p) U; // default-initialisation: nothing for a double
    }
    template <class U, class... Args>
    void construct(U *p, Args &&... args)
    {
        ::new ((void *)p) U(std::forward<Args>(args)...);
    }
};

class Matrix
{
public:
    Matrix()
        : m_nr(0)
        , m_nc(0)
    {
    }
    Matrix(int nr, int nc)
        : m_nr(nr)
        , m_nc(nc)
        , m_d((size_t)nr * nc, 0.0)
    {
    }
    virtual ~Matrix()
    {
    }

    int Nrows() const
    {
        return m_nr;
    }
    int Ncols() const
    {
        return m_nc;
    }
    int Storage() const
    {
        return (int)m_d.size();
    }
    void ReSize(int nr, int nc)
    {
        m_nr = nr;
        m_nc = nc;
        m_d.assign((size_t)nr * nc, 0.0);
    }
    void ReSize(const Matrix &like)
    {
        ReSize(like.m_nr, like.m_nc);
    }
    /** ReSize whose elements are NOT set (NEWMAT's own ReSize leaves them undefined too): for images that are filled
     * completely straight away. Not part of NEWMAT. */
    void ReSizeNoInit(int nr, int nc)
    {
        m_nr = nr;
        m_nc = nc;
        m_d.clear();
        m_d.resize((size_t)nr * nc);
    }
    /** exchange the contents with another matrix (no copy). Not part of NEWMAT. */
    void SwapContents(Matrix &other)
    {
        std::swap(m_nr, other.m_nr);
        std::swap(m_nc, other.m_nc);
        m_d.swap(other.m_d);
    }
    void CleanUp()
    {
        ReSize(0, 0);
    }
    Real *Store()
    {
        return m_d.data();
    }
    const Real *Store() const
    {
        return m_d.data();
    }

    Real &operator()(int i, int j)
    {
        check(i, j);
        return m_d[(size_t)(i - 1) * m_nc + (j - 1)];
    }
    Real operator()(int i, int j) const
    {
        check(i, j);
        return m_d[(size_t)(i - 1) * m_nc + (j - 1)];
    }
    // raw 0-based access without checks (internal use)
    Real &at0(int i, int j)
    {
        return m_d[(size_t)i * m_nc + j];
    }
    Real at0(int i, int j) const
    {
        return m_d[(size_t)i * m_nc + j];
    }

    Matrix &operator=(Real v)
    {
        std::fill(m_d.begin(), m_d.end(), v);
        return *this;
    }
    // "lossy" assignment from another shape (symmetric <- general etc.); here simply a copy
    // with the target's structure re-imposed by the derived class where relevant
    virtual Matrix &operator<<(const Matrix &src)
    {
        assign(src);
        return *this;
    }
    ListFiller operator<<(Real v)
    {
        if (m_d.empty())
            throw IndexException();
        m_d[0] = v;
        return ListFiller(this, 1);
    }
    ListFiller operator<<(int v)
    {
        return operator<<((Real)v);
    }

    // ---- views ----
    MatrixView Row(int i)
    {
        return view(i, 1, 1, m_nc);
    }
    MatrixView Column(int j)
    {
        return view(1, j, m_nr, 1);
    }
    MatrixView Rows(int a, int b)
    {
        return view(a, 1, b - a + 1, m_nc);
    }
    MatrixView Columns(int a, int b)
    {
        return view(1, a, m_nr, b - a + 1);
    }
    MatrixView SubMatrix(int r1, int r2, int c1, int c2)
    {
        return view(r1, c1, r2 - r1 + 1, c2 - c1 + 1);
    }
    MatrixView SymSubMatrix(int a, int b)
    {
        return view(a, a, b - a + 1, b - a + 1);
    }
    Matrix Row(int i) const
    {
        return block(i, 1, 1, m_nc);
    }
    Matrix Column(int j) const
    {
        return block(1, j, m_nr, 1);
    }
    Matrix Rows(int a, int b) const
    {
        return block(a, 1, b - a + 1, m_nc);
    }
    Matrix Columns(int a, int b) const
    {
        return block(1, a, m_nr, b - a + 1);
    }
    Matrix SubMatrix(int r1, int r2, int c1, int c2) const
    {
        return block(r1, c1, r2 - r1 + 1, c2 - c1 + 1);
    }
    Matrix SymSubMatrix(int a, int b) const
    {
        return block(a, a, b - a + 1, b - a + 1);
    }
    Matrix block(int r1, int c1, int nr, int nc) const
    {
        if (r1 < 1 || c1 < 1 || nr < 0 || nc < 0 || r1 + nr - 1 > m_nr || c1 + nc - 1 > m_nc)
            throw IndexException();
        Matrix out(nr, nc);
        for (int i = 0; i < nr; i++)
            for (int j = 0; j < nc; j++)
                out.at0(i, j) = at0(r1 - 1 + i, c1 - 1 + j);
        return out;
    }

    // ---- whole-matrix operations ----
    Matrix t() const
    {
        Matrix out(m_nc, m_nr);
        for (int i = 0; i < m_nr; i++)
            for (int j = 0; j < m_nc; j++)
                out.at0(j, i) = at0(i, j);
        return out;
    }
    Matrix AsRow() const
    {
        Matrix out(1, m_nr * m_nc);
        out.m_d = m_d;
        return out;
    }
    Matrix AsColumn() const
    {
        Matrix out(m_nr * m_nc, 1);
        out.m_d = m_d;
        return out;
    }
    Real AsScalar() const
    {
        if (m_d.size() != 1)
            throw IncompatibleDimensionsException();
        return m_d[0];
    }
    Real Trace() const
    {
        if (m_nr != m_nc)
            throw IncompatibleDimensionsException();
        Real s = 0;
        for (int i = 0; i < m_nr; i++)
            s += at0(i, i);
        return s;
    }
    Real Sum() const
    {
        Real s = 0;
        for (size_t k = 0; k < m_d.size(); k++)
            s += m_d[k];
        return s;
    }
    Real SumSquare() const
    {
        Real s = 0;
        for (size_t k = 0; k < m_d.size(); k++)
            s += m_d[k] * m_d[k];
        return s;
    }
    Real SumAbsoluteValue() const
    {
        Real s = 0;
        for (size_t k = 0; k < m_d.size(); k++)
            s += std::fabs(m_d[k]);
        return s;
    }
    Real Maximum() const
    {
        if (m_d.empty())
            throw IndexException();
        return *std::max_element(m_d.begin(), m_d.end());
    }
    Real Minimum() const
    {
        if (m_d.empty())
            throw IndexException();
        return *std::min_element(m_d.begin(), m_d.end());
    }
    Real MaximumAbsoluteValue() const
    {
        Real s = 0;
        for (size_t k = 0; k < m_d.size(); k++)
            s = std::max(s, std::fabs(m_d[k]));
        return s;
    }
    bool IsZero() const
    {
        for (size_t k = 0; k < m_d.size(); k++)
            if (m_d[k] != 0.0)
                return false;
        return true;
    }

    // LU with partial pivoting; exactly zero pivot = singular
    Matrix i() const
    {
        if (m_nr != m_nc)
            throw IncompatibleDimensionsException();
        const int n = m_nr;
        Matrix lu(*this);
        std::vector<int> piv(n);
        int sign = 1;
        if (!lu.lu_factor(piv, sign))
            throw SingularException();
        Matrix inv(n, n);
        std::vector<Real> col(n);
        for (int j = 0; j < n; j++)
        {
            for (int r = 0; r < n; r++)
                col[r] = (r == j) ? 1.0 : 0.0;
            for (int k = 0; k < n; k++)
                if (piv[k] != k)
                    std::swap(col[k], col[piv[k]]);
            for (int r = 0; r < n; r++)
                for (int k = 0; k < r; k++)
                    col[r] -= lu.at0(r, k) * col[k];
            for (int r = n - 1; r >= 0; r--)
            {
                for (int k = r + 1; k < n; k++)
                    col[r] -= lu.at0(r, k) * col[k];
                col[r] /= lu.at0(r, r);
            }
            for (int r = 0; r < n; r++)
                inv.at0(r, j) = col[r];
        }
        return inv;
    }
    LogAndSign LogDeterminant() const
    {
        if (m_nr != m_nc)
            throw IncompatibleDimensionsException();
        Matrix lu(*this);
        std::vector<int> piv(m_nr);
        int sign = 1;
        if (!lu.lu_factor(piv, sign))
            return LogAndSign(-INFINITY, 0);
        Real l = 0;
        for (int k = 0; k < m_nr; k++)
        {
            Real d = lu.at0(k, k);
            if (d < 0)
            {
                sign = -sign;
                d = -d;
            }
            l += std::log(d);
        }
        return LogAndSign(l, sign);
    }
    Real Determinant() const
    {
        return LogDeterminant().Value();
    }

    // ---- arithmetic ----
    Matrix &operator+=(const Matrix &o)
    {
        same(o);
        for (size_t k = 0; k < m_d.size(); k++)
            m_d[k] += o.m_d[k];
        return *this;
    }
    Matrix &operator-=(const Matrix &o)
    {
        same(o);
        for (size_t k = 0; k < m_d.size(); k++)
            m_d[k] -= o.m_d[k];
        return *this;
    }
    Matrix &operator+=(Real v)
    {
        for (size_t k = 0; k < m_d.size(); k++)
            m_d[k] += v;
        return *this;
    }
    Matrix &operator-=(Real v)
    {
        return operator+=(-v);
    }
    Matrix &operator*=(Real v)
    {
        for (size_t k = 0; k < m_d.size(); k++)
            m_d[k] *= v;
        return *this;
    }
    Matrix &operator/=(Real v)
    {
        for (size_t k = 0; k < m_d.size(); k++)
            m_d[k] /= v;
        return *this;
    }
    // vertical (&) and horizontal (|) concatenation
    Matrix &operator&=(const Matrix &o)
    {
        *this = stack(*this, o);
        return *this;
    }
    Matrix &operator|=(const Matrix &o)
    {
        *this = append(*this, o);
        return *this;
    }
    static Matrix stack(const Matrix &a, const Matrix &b)
    {
        if (a.m_d.empty() && a.m_nc == 0)
            return b;
        if (a.m_nc != b.m_nc)
            throw IncompatibleDimensionsException();
        Matrix out(a.m_nr + b.m_nr, a.m_nc);
        std::copy(a.m_d.begin(), a.m_d.end(), out.m_d.begin());
        std::copy(b.m_d.begin(), b.m_d.end(), out.m_d.begin() + a.m_d.size());
        return out;
    }
    static Matrix append(const Matrix &a, const Matrix &b)
    {
        if (a.m_d.empty() && a.m_nr == 0)
            return b;
        if (a.m_nr != b.m_nr)
            throw IncompatibleDimensionsException();
        Matrix out(a.m_nr, a.m_nc + b.m_nc);
        for (int i = 0; i < a.m_nr; i++)
        {
            for (int j = 0; j < a.m_nc; j++)
                out.at0(i, j) = a.at0(i, j);
            for (int j = 0; j < b.m_nc; j++)
                out.at0(i, a.m_nc + j) = b.at0(i, j);
        }
        return out;
    }

    void assign(const Matrix &src)
    {
        m_nr = src.m_nr;
        m_nc = src.m_nc;
        m_d = src.m_d;
    }
    void set_list_element(int pos, Real v)
    {
        if (pos < 0 || pos >= (int)m_d.size())
            throw IndexException();
        m_d[pos] = v;
    }

protected:
    int m_nr, m_nc;
    std::vector<Real, DefaultInitAllocator<Real> > m_d;

    void check(int i, int j) const
    {
        if (i < 1 || j < 1 || i > m_nr || j > m_nc)
            throw IndexException();
    }
    void same(const Matrix &o) const
    {
        if (o.m_nr != m_nr || o.m_nc != m_nc)
            throw IncompatibleDimensionsException();
    }
    MatrixView view(int r1, int c1, int nr, int nc)
    {
        if (r1 < 1 || c1 < 1 || nr < 0 || nc < 0 || r1 + nr - 1 > m_nr || c1 + nc - 1 > m_nc)
            throw IndexException();
        return MatrixView(this, r1 - 1, c1 - 1, nr, nc);
    }
    bool lu_factor(std::vector<int> &piv, int &sign)
    {
        const int n = m_nr;
        for (int k = 0; k < n; k++)
        {
            int p = k;
            Real best = std::fabs(at0(k, k));
            for (int r = k + 1; r < n; r++)
                if (std::fabs(at0(r, k)) > best)
                {
                    best = std::fabs(at0(r, k));
                    p = r;
                }
            piv[k] = p;
            if (p != k)
            {
                for (int c = 0; c < n; c++)
                    std::swap(at0(k, c), at0(p, c));
                sign = -sign;
            }
            if (at0(k, k) == 0.0)
                return false;
            for (int r = k + 1; r < n; r++)
            {
                at0(r, k) /= at0(k, k);
                const Real f = at0(r, k);
                for (int c = k + 1; c < n; c++)
                    at0(r, c) -= f * at0(k, c);
            }
        }
        return true;
    }
};

typedef Matrix ReturnMatrix;

inline ListFiller ListFiller::operator<<(Real v)
{
    m_m->set_list_element(m_pos, v);
    return ListFiller(m_m, m_pos + 1);
}

// ---- free operators ---------------------------------------------------------------------------
inline Matrix operator+(const Matrix &a, const Matrix &b)
{
    Matrix r(a);
    r += b;
    return r;
}
inline Matrix operator-(const Matrix &a, const Matrix &b)
{
    Matrix r(a);
    r -= b;
    return r;
}
inline Matrix operator-(const Matrix &a)
{
    Matrix r(a);
    r *= -1.0;
    return r;
}
inline Matrix operator+(const Matrix &a, Real v)
{
    Matrix r(a);
    r += v;
    return r;
}
inline Matrix operator-(const Matrix &a, Real v)
{
    Matrix r(a);
    r -= v;
    return r;
}
inline Matrix operator+(Real v, const Matrix &a)
{
    return a + v;
}
inline Matrix operator*(const Matrix &a, Real v)
{
    Matrix r(a);
    r *= v;
    return r;
}
inline Matrix operator*(Real v, const Matrix &a)
{
    return a * v;
}
inline Matrix operator/(const Matrix &a, Real v)
{
    Matrix r(a);
    r /= v;
    return r;
}
inline Matrix operator*(const Matrix &a, const Matrix &b)
{
    if (a.Ncols() != b.Nrows())
        throw IncompatibleDimensionsException();
    Matrix r(a.Nrows(), b.Ncols());
    for (int i = 0; i < a.Nrows(); i++)
        for (int k = 0; k < a.Ncols(); k++)
        {
            const Real v = a.at0(i, k);
            if (v == 0.0)
                continue;
            for (int j = 0; j < b.Ncols(); j++)
                r.at0(i, j) += v * b.at0(k, j);
        }
    return r;
}
inline Matrix operator&(const Matrix &a, const Matrix &b)
{
    return Matrix::stack(a, b);
}
inline Matrix operator|(const Matrix &a, const Matrix &b)
{
    return Matrix::append(a, b);
}
inline bool operator==(const Matrix &a, const Matrix &b)
{
    if (a.Nrows() != b.Nrows() || a.Ncols() != b.Ncols())
        return false;
    for (int i = 0; i < a.Nrows(); i++)
        for (int j = 0; j < a.Ncols(); j++)
            if (a.at0(i, j) != b.at0(i, j)) // NaN != NaN, as the reference's "x == x" checks rely on
                return false;
    return true;
}
inline bool operator!=(const Matrix &a, const Matrix &b)
{
    return !(a == b);
}
inline std::ostream &operator<<(std::ostream &os, const Matrix &m)
{
    for (int i = 0; i < m.Nrows(); i++)
    {
        for (int j = 0; j < m.Ncols(); j++)
            os << m.at0(i, j) << " ";
        os << "\n";
    }
    return os;
}

// ---- derived shapes ---------------------------------------------------------------------------
class ColumnVector : public Matrix
{
public:
    ColumnVector()
    {
    }
    explicit ColumnVector(int n)
        : Matrix(n, 1)
    {
    }
    ColumnVector(const Matrix &m)
    {
        *this = m;
    }
    ColumnVector(const MatrixView &v)
    {
        *this = v.AsMatrix();
    }
    ColumnVector &operator=(const Matrix &m)
    {
        if (m.Ncols() != 1 && m.Nrows() * m.Ncols() != 0 && m.Nrows() != 1)
            throw IncompatibleDimensionsException();
        Matrix::assign(m);
        m_nr = m.Nrows() * m.Ncols();
        m_nc = m_nr ? 1 : 0;
        if (m_nr == 0)
            m_nc = 1;
        return *this;
    }
    ColumnVector &operator=(Real v)
    {
        Matrix::operator=(v);
        return *this;
    }
    ColumnVector &operator=(const MatrixView &v)
    {
        return *this = v.AsMatrix();
    }
    using Matrix::operator<<;
    using Matrix::ReSize;
    void ReSize(int n)
    {
        Matrix::ReSize(n, 1);
    }
    using Matrix::operator();
    Real &operator()(int i)
    {
        if (i < 1 || i > m_nr)
            throw IndexException();
        return m_d[i - 1];
    }
    Real operator()(int i) const
    {
        if (i < 1 || i > m_nr)
            throw IndexException();
        return m_d[i - 1];
    }
    MatrixView Rows(int a, int b)
    {
        return Matrix::Rows(a, b);
    }
    ColumnVector Rows(int a, int b) const
    {
        return ColumnVector(Matrix::Rows(a, b));
    }
};

class RowVector : public Matrix
{
public:
    RowVector()
    {
    }
    explicit RowVector(int n)
        : Matrix(1, n)
    {
    }
    RowVector(const Matrix &m)
    {
        *this = m;
    }
    RowVector(const MatrixView &v)
    {
        *this = v.AsMatrix();
    }
    RowVector &operator=(const Matrix &m)
    {
        if (m.Nrows() != 1 && m.Nrows() * m.Ncols() != 0 && m.Ncols() != 1)
            throw IncompatibleDimensionsException();
        Matrix::assign(m);
        m_nc = m.Nrows() * m.Ncols();
        m_nr = 1;
        return *this;
    }
    RowVector &operator=(Real v)
    {
        Matrix::operator=(v);
        return *this;
    }
    using Matrix::operator<<;
    using Matrix::ReSize;
    void ReSize(int n)
    {
        Matrix::ReSize(1, n);
    }
    using Matrix::operator();
    Real &operator()(int i)
    {
        if (i < 1 || i > m_nc)
            throw IndexException();
        return m_d[i - 1];
    }
    Real operator()(int i) const
    {
        if (i < 1 || i > m_nc)
            throw IndexException();
        return m_d[i - 1];
    }
};

// Stored dense; kept symmetric by construction (lower triangle wins on lossy assignment, as
// NEWMAT's "sym << general" does).
class SymmetricMatrix : public Matrix
{
public:
    SymmetricMatrix()
    {
    }
    explicit SymmetricMatrix(int n)
        : Matrix(n, n)
    {
    }
    SymmetricMatrix(const Matrix &m)
    {
        take_lower(m);
    }
    SymmetricMatrix &operator=(const Matrix &m)
    {
        take_lower(m);
        return *this;
    }
    SymmetricMatrix &operator=(Real v)
    {
        Matrix::operator=(v);
        return *this;
    }
    Matrix &operator<<(const Matrix &m)
    {
        take_lower(m);
        return *this;
    }
    using Matrix::operator<<;
    using Matrix::ReSize;
    void ReSize(int n)
    {
        Matrix::ReSize(n, n);
    }
    // element proxy keeping both triangles in step
    class Ref
    {
    public:
        Ref(SymmetricMatrix *m, int i, int j)
            : m_m(m)
            , m_i(i)
            , m_j(j)
        {
        }
        operator Real() const
        {
            return m_m->at0(m_i, m_j);
        }
        Ref &operator=(Real v)
        {
            m_m->at0(m_i, m_j) = v;
            m_m->at0(m_j, m_i) = v;
            return *this;
        }
        Ref &operator=(const Ref &o)
        {
            return operator=((Real)o);
        }
        Ref &operator+=(Real v)
        {
            return operator=((Real)*this + v);
        }
        Ref &operator-=(Real v)
        {
            return operator=((Real)*this - v);
        }
        Ref &operator*=(Real v)
        {
            return operator=((Real)*this * v);
        }

    private:
        SymmetricMatrix *m_m;
        int m_i, m_j;
    };
    Ref operator()(int i, int j)
    {
        check(i, j);
        return Ref(this, i - 1, j - 1);
    }
    Real operator()(int i, int j) const
    {
        check(i, j);
        return at0(i - 1, j - 1);
    }
    SymmetricMatrix i() const
    {
        return SymmetricMatrix(Matrix::i());
    }

private:
    void take_lower(const Matrix &m)
    {
        if (m.Nrows() != m.Ncols())
            throw IncompatibleDimensionsException();
        Matrix::assign(m);
        for (int i = 0; i < m_nr; i++)
            for (int j = 0; j < i; j++)
                at0(j, i) = at0(i, j);
    }
};

class DiagonalMatrix : public Matrix
{
public:
    DiagonalMatrix()
    {
    }
    explicit DiagonalMatrix(int n)
        : Matrix(n, n)
    {
    }
    DiagonalMatrix(const Matrix &m)
    {
        take_diag(m);
    }
    DiagonalMatrix &operator=(const Matrix &m)
    {
        take_diag(m);
        return *this;
    }
    DiagonalMatrix &operator=(Real v)
    {
        for (int i = 0; i < m_nr; i++)
            at0(i, i) = v;
        return *this;
    }
    Matrix &operator<<(const Matrix &m)
    {
        take_diag(m);
        return *this;
    }
    using Matrix::ReSize;
    void ReSize(int n)
    {
        Matrix::ReSize(n, n);
    }
    using Matrix::operator();
    Real &operator()(int i)
    {
        check(i, i);
        return at0(i - 1, i - 1);
    }
    Real operator()(int i) const
    {
        check(i, i);
        return at0(i - 1, i - 1);
    }
    DiagonalMatrix i() const
    {
        DiagonalMatrix r(m_nr);
        for (int k = 0; k < m_nr; k++)
        {
            if (at0(k, k) == 0.0)
                throw SingularException();
            r.at0(k, k) = 1.0 / at0(k, k);
        }
        return r;
    }

private:
    void take_diag(const Matrix &m)
    {
        if (m.Nrows() != m.Ncols())
            throw IncompatibleDimensionsException();
        Matrix::ReSize(m.Nrows(), m.Nrows());
        for (int i = 0; i < m_nr; i++)
            at0(i, i) = m.at0(i, i);
    }
};

class IdentityMatrix : public DiagonalMatrix
{
public:
    explicit IdentityMatrix(int n = 0)
        : DiagonalMatrix(n)
    {
        for (int i = 0; i < n; i++)
            at0(i, i) = 1.0;
    }
};

// ---- MatrixView implementation ------------------------------------------------------------------
inline MatrixView &MatrixView::operator=(const Matrix &src)
{
    if (src.Nrows() * src.Ncols() != m_nr * m_nc)
        throw IncompatibleDimensionsException();
    // accept row/column orientation mismatches for vectors
    const Real *s = src.Store();
    int k = 0;
    for (int i = 0; i < m_nr; i++)
        for (int j = 0; j < m_nc; j++)
            m_m->at0(m_r0 + i, m_c0 + j) = s[k++];
    return *this;
}
inline MatrixView &MatrixView::operator=(const MatrixView &src)
{
    return operator=(src.AsMatrix());
}
inline MatrixView &MatrixView::operator=(Real v)
{
    for (int i = 0; i < m_nr; i++)
        for (int j = 0; j < m_nc; j++)
            m_m->at0(m_r0 + i, m_c0 + j) = v;
    return *this;
}
inline MatrixView &MatrixView::operator<<(const Matrix &src)
{
    return operator=(src);
}
inline Matrix MatrixView::AsMatrix() const
{
    Matrix out(m_nr, m_nc);
    for (int i = 0; i < m_nr; i++)
        for (int j = 0; j < m_nc; j++)
            out.at0(i, j) = m_m->at0(m_r0 + i, m_c0 + j);
    return out;
}
inline MatrixView::operator Matrix() const
{
    return AsMatrix();
}
inline Real MatrixView::operator()(int i, int j) const
{
    if (i < 1 || j < 1 || i > m_nr || j > m_nc)
        throw IndexException();
    return m_m->at0(m_r0 + i - 1, m_c0 + j - 1);
}
inline Real MatrixView::operator()(int i) const
{
    if (m_nc == 1)
        return operator()(i, 1);
    return operator()(1, i);
}
inline Matrix MatrixView::t() const
{
    return AsMatrix().t();
}
inline Real MatrixView::Sum() const
{
    return AsMatrix().Sum();
}
inline Real MatrixView::Maximum() const
{
    return AsMatrix().Maximum();
}
inline Real MatrixView::Minimum() const
{
    return AsMatrix().Minimum();
}
inline Real MatrixView::AsScalar() const
{
    return AsMatrix().AsScalar();
}
inline Matrix MatrixView::Rows(int a, int b) const
{
    return AsMatrix().block(a, 1, b - a + 1, m_nc);
}
inline Matrix MatrixView::Columns(int a, int b) const
{
    return AsMatrix().block(1, a, m_nr, b - a + 1);
}
inline bool MatrixView::IsZero() const
{
    return AsMatrix().IsZero();
}
inline Matrix operator+(const MatrixView &a, const Matrix &b)
{
    return a.AsMatrix() + b;
}
inline Matrix operator-(const MatrixView &a, const Matrix &b)
{
    return a.AsMatrix() - b;
}
inline Matrix operator-(const MatrixView &a, const MatrixView &b)
{
    return a.AsMatrix() - b.AsMatrix();
}
inline Matrix operator+(const MatrixView &a, const MatrixView &b)
{
    return a.AsMatrix() + b.AsMatrix();
}
inline Matrix operator*(const MatrixView &a, const Matrix &b)
{
    return a.AsMatrix() * b;
}
inline Matrix operator*(const MatrixView &a, Real v)
{
    return a.AsMatrix() * v;
}
inline Matrix operator/(const MatrixView &a, Real v)
{
    return a.AsMatrix() / v;
}
inline std::ostream &operator<<(std::ostream &os, const MatrixView &v)
{
    return os << v.AsMatrix();
}

} // namespace NEWMAT

// FSL's NEWMAT headers bring namespace std into scope for everything that includes them, and
// model sources written against them rely on it (e.g. examples/exp_models.cc:26 uses an
// unqualified `string`). Define FABBER_NEWMAT_NO_USING_STD to opt out.
#ifndef FABBER_NEWMAT_NO_USING_STD
using namespace std;
#endif
