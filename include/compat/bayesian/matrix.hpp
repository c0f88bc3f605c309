// compat/bayesian/matrix.hpp -- this repository's own, API-compatible stand-in for the
// reference's bn::matrix_type (bayesian/matrix.hpp:10-159), so that the drop-in inference headers
// can be built and tested where the reference is not installed (the GPU box).  Storage is one
// std::vector<double> per row, as the public signature `std::vector<double>& operator[](size_t)`
// requires; test scaffolding, never on the hot path (the C ABI moves flat arrays).
#ifndef BNI_MATRIX_HPP
#define BNI_MATRIX_HPP

#include <cassert>
#include <cstddef>
#include <iterator>
#include <vector>

namespace bn {

class matrix_type {
public:
    matrix_type() = default;
    matrix_type(std::size_t const height, std::size_t const width, double const default_value = 0.0)
    {
        resize(height, width, default_value);
    }
    virtual ~matrix_type() = default;

    std::size_t height() const { return rows_.size(); }
    std::size_t width() const { return width_; }

    // grow / shrink keeping the overlapping part, new cells = default_value
    void resize(std::size_t const height, std::size_t const width, double const default_value = 0.0)
    {
        for(auto& r : rows_) r.resize(width, default_value);
        rows_.resize(height, std::vector<double>(width, default_value));
        width_ = width;
    }

    // fills row by row from the range; false when the range is too short
    template<class InputIterator>
    bool assign(InputIterator begin, InputIterator const& end)
    {
        if(static_cast<std::size_t>(std::distance(begin, end)) < height() * width_) return false;
        for(auto& r : rows_)
            for(auto& x : r) x = *begin++;
        return true;
    }

    std::vector<double>& operator[](std::size_t const y) { return rows_[y]; }
    std::vector<double> const& operator[](std::size_t const y) const { return rows_[y]; }

    // element-wise product
    matrix_type& operator%=(matrix_type const& rhs)
    {
        assert(width() == rhs.width() && height() == rhs.height());
        for(std::size_t y = 0; y < height(); ++y)
            for(std::size_t x = 0; x < width_; ++x) rows_[y][x] *= rhs.rows_[y][x];
        return *this;
    }
    matrix_type operator%(matrix_type const& rhs) const
    {
        matrix_type out(*this);
        out %= rhs;
        return out;
    }

    // matrix product
    matrix_type& operator*=(matrix_type const& rhs)
    {
        assert(width() == rhs.height());
        matrix_type out(height(), rhs.width(), 0.0);
        for(std::size_t y = 0; y < height(); ++y)
            for(std::size_t x = 0; x < rhs.width(); ++x)
            {
                double acc = 0.0;
                for(std::size_t t = 0; t < width_; ++t) acc += rows_[y][t] * rhs.rows_[t][x];
                out.rows_[y][x] = acc;
            }
        *this = out;
        return *this;
    }
    matrix_type operator*(matrix_type const& rhs) const
    {
        matrix_type out(*this);
        out *= rhs;
        return out;
    }

private:
    std::size_t width_ = 0;
    std::vector<std::vector<double>> rows_;
};

} // namespace bn

using bn::matrix_type;

template<class T>
bn::matrix_type operator*(bn::matrix_type const& m, T const& scalar)
{
    bn::matrix_type out(m);
    for(std::size_t y = 0; y < out.height(); ++y)
        for(auto& x : out[y]) x *= scalar;
    return out;
}

template<class T>
bn::matrix_type operator*(T const& scalar, bn::matrix_type const& m)
{
    return m * scalar;
}

#endif // BNI_MATRIX_HPP
