// API-shape declaration of boost::shared_ptr (what pcl::PointCloud<T>::Ptr is in PCL 1.7).  TEST-ONLY.
#ifndef AGH_TEST_STUB_BOOST_SHARED_PTR
#define AGH_TEST_STUB_BOOST_SHARED_PTR
namespace boost
{
template <class T>
class shared_ptr
{
public:
  shared_ptr();
  template <class Y>
  explicit shared_ptr(Y* p);
  shared_ptr(const shared_ptr&);
  shared_ptr& operator=(const shared_ptr&);
  void reset();
  template <class Y>
  void reset(Y* p);
  T& operator*() const;
  T* operator->() const;
  T* get() const;
  explicit operator bool() const;
};
}  // namespace boost
#endif
